#!/bin/bash
# (TA_* / TCP_* counter sets are deliberately absent: one pass with them ran for 15 minutes on this pool.)
# PMC counters of the split (LDS-DMA) conv kernel:  bash tools/gpu_pmc2.sh "<layer> <tile>" ...
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "$@"; do
  set -- $cfg; layer=$1; tile=$2
  python tools/conv_micro.py --layer $layer --tile $tile --precision bf16x3 --split planes --reps 10 2>&1 | grep -v amdgpu
  (cd /tmp
   i=0
   for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
              "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
              "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
     i=$((i+1))
     rm -rf "$GRAFT_REPO_ROOT/gpurun_out/q_$i"
     timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/q_$i" -o pmc -- python "$GRAFT_REPO_ROOT/tools/conv_micro.py" --layer $layer --tile $tile --precision bf16x3 --split planes --reps 5 > /dev/null 2>&1
   done)
  python - <<'PY'
import csv, glob, collections
out = {}
for f in sorted(glob.glob("gpurun_out/q_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "conv_split" in r.get("Kernel_Name", ""):
            a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, v in agg.items(): out[k] = v[1] / max(v[0], 1)
print({k: round(v / 1e6, 2) for k, v in out.items()})
wc = out.get("SQ_WAVE_CYCLES", 1)
g = out.get("GRBM_GUI_ACTIVE", 1)
print("per-wave: active %.0f%% (vmem %.0f%% lds %.0f%% valu %.0f%% sca %.0f%%) wait_any %.0f%% wait_inst %.0f%% | mfma busy %.1f%%" % (
    100 * out.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * out.get("SQ_ACTIVE_INST_VMEM", 0) / wc, 100 * out.get("SQ_ACTIVE_INST_LDS", 0) / wc,
    100 * out.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * out.get("SQ_ACTIVE_INST_SCA", 0) / wc,
    100 * out.get("SQ_WAIT_ANY", 0) / wc, 100 * out.get("SQ_WAIT_INST_ANY", 0) / wc,
    100 * out.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (g / 8 * 1024)))
PY
done
