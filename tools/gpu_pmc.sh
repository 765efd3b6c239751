#!/bin/bash
# PMC counters of one conv layer / tile:  bash tools/gpu_pmc.sh "<layer> <tile> <precision>" ...
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "$@"; do
  set -- $cfg; layer=$1; tile=$2; prec=$3
  python tools/conv_micro.py --layer $layer --tile $tile --precision $prec --reps 10
  (cd /tmp
   for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_INSTS_SALU"; do
     tag=$(echo $set | cut -d' ' -f1)
     rm -rf "$GRAFT_REPO_ROOT/gpurun_out/p_$tag"
     timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/p_$tag" -o pmc -- python "$GRAFT_REPO_ROOT/tools/conv_micro.py" --layer $layer --tile $tile --precision $prec --reps 5 > /dev/null 2>&1
   done)
  python - <<'PY'
import csv, glob, collections
out = {}
for f in sorted(glob.glob("gpurun_out/p_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "conv_igemm" in r.get("Kernel_Name", ""):
            a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, v in agg.items(): out[k] = v[1] / max(v[0], 1)
wc = out["SQ_WAVE_CYCLES"]
print({k: round(v / 1e6, 2) for k, v in out.items()})
print("per-wave: active %.0f%% wait_any %.0f%% wait_inst %.0f%% | mfma busy %.1f%% | VALU/MFMA %.1f | LDS conflict %.0f%% of LDS active" % (
    100 * out["SQ_ACTIVE_INST_ANY"] / wc, 100 * out["SQ_WAIT_ANY"] / wc, 100 * out["SQ_WAIT_INST_ANY"] / wc,
    100 * out["SQ_VALU_MFMA_BUSY_CYCLES"] / (out["GRBM_GUI_ACTIVE"] / 8 * 1024), out["SQ_INSTS_VALU"] / out["SQ_INSTS_MFMA"],
    100 * out["SQ_LDS_BANK_CONFLICT"] / max(out["SQ_LDS_IDX_ACTIVE"], 1)))
PY
done
