#!/bin/bash
# round-5 visit A: fold-convolution tests, the suites its change touches, a short bench with and without it
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_foldconv_gpu.py -m gpu -q -x -rA -p no:cacheprovider > gpurun_out/r5a_foldconv.log 2>&1; echo "foldconv exit $?"
grep -E "passed|failed|error" gpurun_out/r5a_foldconv.log | tail -3; grep -E "^FAILED|^ERROR|Error|assert" gpurun_out/r5a_foldconv.log | head -20
timeout 900 python -m pytest tests/test_taps_gpu.py tests/test_fgt_gpu.py tests/test_clip_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/r5a_suites.log 2>&1; echo "suites exit $?"
grep -E "passed|failed|error" gpurun_out/r5a_suites.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/r5a_suites.log | head -20
for fc in 1 0; do
  FGT_FOLD_CONV=$fc timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-fp32-exact --no-c4 > gpurun_out/r5a_bench_fc$fc.log 2>&1; echo "bench fc=$fc exit $?"
  cp gpurun_out/bench_detail.json gpurun_out/r5a_bench_detail_fc$fc.json
  python - <<PY
import json; d=json.load(open('gpurun_out/bench_detail.json'))
print('fold_conv=$fc', d['value'],'fps', d['ms_per_step'],'ms', 'parity', d.get('parity_vs_cpu_oracle',{}).get('max_abs_diff'), 'enqueue', d.get('host_enqueue_ms_per_step'))
for r in d.get('rooflines',[]): print('  ', r['kind'], r['frac'], r.get('algorithmic_tflops', r.get('achieved')),r['unit'], r['kernel_ms_per_step'],'ms/step', r['avg_launch_us'],'us/launch', r.get('launches'))
PY
done
