#!/bin/bash
# round-5 visit C: the suites the RAFT / epilogue changes touch, then the default bench line
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_foldconv_gpu.py tests/test_taps_gpu.py tests/test_flow_gpu.py tests/test_fgt_gpu.py tests/test_split_gpu.py -m gpu -q -rA -p no:cacheprovider > gpurun_out/r5c_suites.log 2>&1; echo "suites exit $?"
grep -E "passed|failed|error" gpurun_out/r5c_suites.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/r5c_suites.log | head -20
grep "\[parity\] RAFT" gpurun_out/r5c_suites.log | cut -c1-200
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/r5c_bench.log 2>&1; echo "bench exit $?"
cp gpurun_out/bench_detail.json gpurun_out/r5c_bench_detail.json
tail -1 gpurun_out/r5c_bench.log > gpurun_out/r5c_bench_line.json; wc -c gpurun_out/r5c_bench_line.json
python - <<PY
import json; d=json.load(open('gpurun_out/bench_detail.json'))
print(d['value'],'fps', d['ms_per_step'],'ms', 'parity', d.get('parity_vs_cpu_oracle',{}).get('max_abs_diff'), 'enqueue', d.get('host_enqueue_ms_per_step'), 'probe us', d.get('host_launch_us_probe'), d['config'].get('hip_graphs'), d['config'].get('graph_probe'))
for r in d.get('rooflines',[]): print('  ', r['kind'], r['frac'], r.get('algorithmic_tflops', r.get('achieved')),r['unit'], r['kernel_ms_per_step'],'ms/step', r['avg_launch_us'],'us/launch', r.get('launches'))
c4=d.get('c4',{})
print('c4', {k:{kk:vv for kk,vv in v.items() if kk.startswith('ms_per')} for k,v in c4.get('stages',{}).items()}, (c4.get('pipeline_frames_per_s') or {}).get('value'), c4.get('error'))
print('c2', c4.get('c2_spatial_mhsa'))
f=d.get('fp32_exact'); print('fp32', f and (f['value'], f['ms_per_step']))
print('cpu', d.get('cpu_baseline'))
PY
