#!/bin/bash
# round 3, visit D: RAFT / LAFC split chains (parity + timing)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_flow_gpu.py tests/test_c1_plumbing.py tests/test_split_gpu.py -m gpu -q -rA -p no:cacheprovider -k "not wide_tiles" > gpurun_out/pytest_d.log 2>&1
echo "pytest exit: $?"; grep -E "passed|failed|error" gpurun_out/pytest_d.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/pytest_d.log | head; grep "\[parity\]" gpurun_out/pytest_d.log | grep -E "RAFT|raft|lafc" | cut -c1-250 | head -20
echo "== RAFT breakdown"
timeout 600 python tools/raft_breakdown.py > gpurun_out/raft_breakdown.txt 2>&1; cut -c1-200 gpurun_out/raft_breakdown.txt | head -30
echo "== bench (c4 only matters)"
timeout 900 python bench.py --steps 3 --warmup 1 --no-fp32-exact --no-f16 --no-cpu-baseline > gpurun_out/bench_d.log 2>&1; echo "bench exit: $?"
grep '^{' gpurun_out/bench_d.log > gpurun_out/bench_d.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_d.json'))
print(d['value'],'fps', d['ms_per_step'],'ms')
c=d.get('c4',{})
if 'error' in c: print(c)
for k,v in c.get('stages',{}).items(): print(k, {a:b for a,b in v.items() if a not in ('roofline','pipeline','note','solver')}, v.get('roofline',{}).get('frac'))
print(c.get('pipeline_frames_per_s',{}).get('value'), c.get('pipeline_frames_per_s',{}).get('stages_ms'))
for r in c.get('rooflines',[]): print(r)
PY
