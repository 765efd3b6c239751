#!/bin/bash
# round-5 visit J: chunk-major tap kernels as the only order: tap / fold / model suites, bench
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_taps_gpu.py tests/test_foldconv_gpu.py tests/test_fgt_gpu.py tests/test_clip_gpu.py tests/test_flow_gpu.py tests/test_split_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/r5j_suites.log 2>&1; echo "suites exit $?"
grep -E "passed|failed|error" gpurun_out/r5j_suites.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/r5j_suites.log | head -20
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-fp32-exact > gpurun_out/r5j_bench.log 2>&1; echo "bench exit $?"
cp gpurun_out/bench_detail.json gpurun_out/r5j_bench_detail.json
python - <<PY
import json; d=json.load(open('gpurun_out/bench_detail.json'))
print(d['value'],'fps', d['ms_per_step'],'ms', 'enqueue', d.get('host_enqueue_ms_per_step'))
for r in d.get('rooflines',[]): print('  ', r['kind'], r['frac'], r.get('algorithmic_tflops', r.get('achieved')),r['unit'], r['kernel_ms_per_step'],'ms/step', r.get('traffic_over_algorithmic'))
c4=d.get('c4',{})
print('c4', {k:{kk:vv for kk,vv in v.items() if kk.startswith('ms_per')} for k,v in c4.get('stages',{}).items()}, (c4.get('pipeline_frames_per_s') or {}).get('value'), c4.get('error'))
PY
timeout 280 python tools/conv_breakdown.py > gpurun_out/r5j_conv_breakdown.txt 2>&1; head -12 gpurun_out/r5j_conv_breakdown.txt | cut -c1-150
