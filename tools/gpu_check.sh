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
echo "== bench (variants)"
for v in "--precision fp32 --no-cache" "--precision bf16x3"; do
  tag=$(echo "$v" | tr -d ' -')
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $v > gpurun_out/bench_$tag.log 2>&1; echo "bench $v exit: $?"
  grep '^{' gpurun_out/bench_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'],'fps', d['ms_per_step'],'ms', d.get('roofline'))"
done
echo "== bench (default, with cpu baseline)"
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/bench.log 2>&1; echo "bench exit: $?" | tee -a gpurun_out/bench.log
tail -3 gpurun_out/bench.log
echo "== tune_conv"
for pipe in 0 1; do
FGT_AUTOTUNE=0 FGT_CONV_PIPE=$pipe FGT_CONV_PRECISION=bf16x3 timeout 600 python tools/tune_conv.py --t 17 --tiles 128x128,64x64,128x64,256x128 > gpurun_out/tune_conv_bf16x3_pipe$pipe.log 2>&1; tail -32 gpurun_out/tune_conv_bf16x3_pipe$pipe.log
done
FGT_AUTOTUNE=0 FGT_CONV_PRECISION=fp32 timeout 600 python tools/tune_conv.py --t 17 --tiles 128x128,64x64 > gpurun_out/tune_conv_fp32.log 2>&1; tail -32 gpurun_out/tune_conv_fp32.log
if [ "${1:-}" != "quick" ]; then
  echo "== rocprofv3 kernel stats"
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o fgt -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-prof --precision bf16x3 > "$GRAFT_REPO_ROOT/gpurun_out/rocprof.log" 2>&1)
  echo "rocprof exit: $?"
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cut -c1-160 "$f" | head -16
fi
