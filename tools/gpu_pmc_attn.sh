#!/bin/bash
# SQ counters of the temporal attention kernel (bf16x3, t = 17):  bash tools/gpu_pmc_attn.sh
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/attn_micro.py --precision bf16x3 --t 17 2>&1 | grep attention
(cd /tmp
 i=0
 for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" \
            "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES"; do
   i=$((i+1)); rm -rf "$GRAFT_REPO_ROOT/gpurun_out/a_$i"
   timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/a_$i" -o pmc -- python "$GRAFT_REPO_ROOT/tools/attn_micro.py" --precision bf16x3 --t 17 --reps 5 > /dev/null 2>&1
 done)
python - <<'PY'
import csv, glob, collections
out = {}
for f in sorted(glob.glob("gpurun_out/a_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "attn" in r.get("Kernel_Name", ""):
            a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, v in agg.items(): out[k] = v[1] / max(v[0], 1)
print({k: round(v / 1e6, 2) for k, v in out.items()})
wc = out.get("SQ_WAVE_CYCLES", 1); g = out.get("GRBM_GUI_ACTIVE", 1)
print("per-wave: active %.0f%% (valu %.0f%% lds %.0f%%) wait_any %.0f%% wait_inst %.0f%% | mfma busy %.1f%% | VALU/MFMA %.1f | LDS conflict %.0f%% of LDS active" % (
    100 * out.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * out.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * out.get("SQ_ACTIVE_INST_LDS", 0) / wc,
    100 * out.get("SQ_WAIT_ANY", 0) / wc, 100 * out.get("SQ_WAIT_INST_ANY", 0) / wc,
    100 * out.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (g / 8 * 1024), out.get("SQ_INSTS_VALU", 0) / max(out.get("SQ_INSTS_MFMA", 1), 1),
    100 * out.get("SQ_LDS_BANK_CONFLICT", 0) / max(out.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
PY
