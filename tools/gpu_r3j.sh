#!/bin/bash
# round 3, visit J: tap-reusing kernel in the product library (geometry routing + autotuned tiles): parity subsets, bench, RAFT breakdown
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_taps_gpu.py tests/test_split_gpu.py tests/test_flow_gpu.py -q -x -p no:cacheprovider > gpurun_out/pytest_j.log 2>&1
echo "pytest exit: $?"; grep -E "passed|failed|error" gpurun_out/pytest_j.log | tail -2; grep -E "^FAILED|^ERROR|Error" gpurun_out/pytest_j.log | head -20
FGT_HIP_LIB=$PWD/fgt_amd/lib/libfgt_hip_diag.so timeout 600 python -m pytest tests/test_taps_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -2
echo "== bench"
timeout 900 python bench.py --steps 5 --warmup 1 --no-fp32-exact --no-f16 > gpurun_out/bench_j.log 2>&1; echo "bench exit: $?"
grep '^{' gpurun_out/bench_j.log > gpurun_out/bench_j.json; tail -2 gpurun_out/bench_j.log | cut -c1-200
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_j.json'))
print(d['value'],'fps', d['ms_per_step'],'ms', 'parity', d.get('parity_vs_cpu_oracle',{}).get('max_abs_diff'))
for r in d.get('rooflines',[])[:3]: print('  ', r['kind'], r['bound'][:4], r['frac'], r['achieved'], r['unit'], r['kernel_ms_per_step'],'ms/step')
c=d.get('c4',{})
if 'error' in c: print(c)
for k,v in c.get('stages',{}).items(): print(k, {a:b for a,b in v.items() if a not in ('roofline','pipeline','note','solver')}, v.get('roofline',{}).get('frac'))
print(c.get('pipeline_frames_per_s',{}).get('value'), c.get('pipeline_frames_per_s',{}).get('stages_ms'))
PY
echo "== RAFT breakdown"
timeout 600 python tools/raft_breakdown.py > gpurun_out/raft_breakdown_j.txt 2>&1; cut -c1-200 gpurun_out/raft_breakdown_j.txt | head -18
