#!/bin/bash
# round-5 visit F: encoder group merge A/B (tests of the model suites first), conv breakdown
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fgt_gpu.py tests/test_clip_gpu.py tests/test_foldconv_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/r5f_suites.log 2>&1; echo "suites exit $?"
grep -E "passed|failed|error" gpurun_out/r5f_suites.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/r5f_suites.log | head -20
for mg in 1 0; do
  FGT_ENC_MERGE_GROUPS=$mg timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-fp32-exact --no-c4 > gpurun_out/r5f_bench_mg$mg.log 2>&1; echo "bench mg=$mg exit $?"
  cp gpurun_out/bench_detail.json gpurun_out/r5f_bench_detail_mg$mg.json
  python - <<PY
import json; d=json.load(open('gpurun_out/bench_detail.json'))
print('merge_groups=$mg', d['value'],'fps', d['ms_per_step'],'ms', 'enqueue', d.get('host_enqueue_ms_per_step'), d.get('host_ms_per_step_in_timed_region_incl_queue_backpressure'))
for r in d.get('rooflines',[])[:4]: print('  ', r['kind'], r['frac'], r.get('algorithmic_tflops', r.get('achieved')),r['unit'], r['kernel_ms_per_step'],'ms/step')
PY
done
timeout 280 python tools/conv_breakdown.py > gpurun_out/r5f_conv_breakdown.txt 2>&1; grep "640\|(20, 60, 108, 256)" gpurun_out/r5f_conv_breakdown.txt | head
# host time in the timed region vs number of steps (queue back-pressure): 3 vs 20 steps, same box
for k in 3 20; do
  timeout 600 python bench.py --steps $k --warmup 1 --no-cpu-baseline --no-fp32-exact --no-c4 > gpurun_out/r5f_bench_steps$k.log 2>&1
  python - <<PY
import json; d=json.load(open('gpurun_out/bench_detail.json'))
print('steps=$k', d['value'],'fps', 'host enqueue (probe, idle queue)', d.get('host_enqueue_ms_per_step'), 'host ms/step in the timed region', d.get('host_ms_per_step_in_timed_region_incl_queue_backpressure'), 'launch probe us', d.get('host_launch_us_probe'))
PY
done
