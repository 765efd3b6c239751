#!/bin/bash
# One GPU-box visit: parity tests, smoke, short bench, rocprofv3 kernel stats.  Everything lands in gpurun_out/.
# usage (from the repo root on the GPU box):  bash tools/gpu_check.sh [quick]
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu" 
timeout 1200 python -m pytest tests -m gpu -q -rA --durations=5 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" | tee -a gpurun_out/pytest_gpu.log
grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -3
echo "== smoke"
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?" | tee -a gpurun_out/smoke.log
tail -2 gpurun_out/smoke.log
echo "== bench (default, with cpu baseline)"
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2>&1; echo "bench exit: $?" | tee -a gpurun_out/bench.log
grep '^{' gpurun_out/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'],'fps', d['ms_per_step'],'ms', d.get('roofline'), d.get('cpu_baseline',{}).get('value'))"
echo "== bench fp32 exact, no cache (reference-equivalent work)"
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision fp32 --no-cache > gpurun_out/bench_fp32_nocache.log 2>&1
grep '^{' gpurun_out/bench_fp32_nocache.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'],'fps', d['ms_per_step'],'ms', d.get('roofline'))"
echo "== tune_conv"
FGT_AUTOTUNE=0 FGT_CONV_PRECISION=bf16x3 timeout 600 python tools/tune_conv.py --t 17 --tiles 128x128,64x64,256x128,128x128x8,256x128x16 > gpurun_out/tune_conv_bf16x3.log 2>&1; tail -30 gpurun_out/tune_conv_bf16x3.log | cut -c1-130
echo "== PMC on one layer (enc8, bf16x3 128x128)"
(cd /tmp && rocprofv3 -L > "$GRAFT_REPO_ROOT/gpurun_out/counters.txt" 2>&1
 for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
   tag=$(echo $set | cut -d' ' -f1)
   timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag" -o pmc -- python "$GRAFT_REPO_ROOT/tools/conv_micro.py" --layer enc8 --tile 128x128 --precision bf16x3 --reps 5 > "$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log" 2>&1
   echo "pmc $tag exit $?"; tail -1 "$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log"
 done)
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "conv_igemm" in r.get("Kernel_Name", ""):
            a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    print(f.split("/")[1], {k: round(v[1] / max(v[0], 1), 1) for k, v in agg.items()})
PY
if [ "${1:-}" != "quick" ]; then
  echo "== flow bench (LAFC / RAFT / config-5 window)"
  timeout 600 python tools/flow_bench.py --cpu > gpurun_out/flow_bench.log 2>&1; tail -5 gpurun_out/flow_bench.log | cut -c1-250
  echo "== PMC HBM traffic of conv_igemm over bench.py"
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && FGT_TUNING_FILE="$GRAFT_REPO_ROOT/gpurun_out/tuning.json" timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmcb_$c" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-prof > "$GRAFT_REPO_ROOT/gpurun_out/pmcb_$c.log" 2>&1)
  done
  python tools/pmc_traffic.py gpurun_out/pmcb_FETCH_SIZE gpurun_out/pmcb_WRITE_SIZE bf16x3 gpurun_out/conv_traffic.json
  echo "== rocprofv3 kernel stats"
  rm -rf gpurun_out/prof
  (cd /tmp && FGT_TUNING_FILE="$GRAFT_REPO_ROOT/gpurun_out/tuning.json" timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o fgt -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-prof --precision bf16x3 > "$GRAFT_REPO_ROOT/gpurun_out/rocprof.log" 2>&1)
  echo "rocprof exit: $?"
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cut -c1-160 "$f" | head -16
fi
if [ "${1:-}" != "quick" ]; then
  echo "== attention kernels (algorithmic TFLOP/s; fp32 peak 157.3, bf16x3 issues 3x)"
  for p in fp32 bf16x3; do python tools/attn_micro.py --precision $p --t 17 2>&1 | grep attention; python tools/attn_micro.py --precision $p --t 17 --spatial 2>&1 | grep attention; done | tee gpurun_out/attn_micro.log
  echo "== bench exact fp32 WITH the feature cache and window batching"
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision fp32 > gpurun_out/bench_fp32.log 2>&1
  grep '^{' gpurun_out/bench_fp32.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'],'fps', d['ms_per_step'],'ms', d.get('roofline'))"
  echo "== 2-rank rehearsal of bench.py on this single GPU (gloo, collectives staged through the host): weak headline + strong extra"
  FGT_BENCH_SHARE_GPU=1 FGT_BENCH_BACKEND=gloo FGT_TUNING_FILE="$PWD/gpurun_out/tuning.json" timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-prof > gpurun_out/bench_2rank_rehearsal.log 2>&1
  echo "rehearsal exit: $?"; grep '^{' gpurun_out/bench_2rank_rehearsal.log | cut -c1-1200
fi
