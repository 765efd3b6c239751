#!/bin/bash
# round 3, visit C: blend on-chip test, RAFT breakdown (32 pairs, 864x480), bench with NT stores + RAFT batch 32
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_blend_pinned.py tests/test_dist_gpu.py -m gpu -q -rA -p no:cacheprovider > gpurun_out/pytest_c.log 2>&1
echo "pytest exit: $?"; grep -E "passed|failed|error" gpurun_out/pytest_c.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/pytest_c.log | head; grep "\[parity\]" gpurun_out/pytest_c.log | grep -E "on-chip|2 ranks" | cut -c1-250
echo "== RAFT breakdown"
timeout 600 python tools/raft_breakdown.py > gpurun_out/raft_breakdown.txt 2>&1; cut -c1-200 gpurun_out/raft_breakdown.txt | head -45
echo "== bench"
timeout 900 python bench.py --steps 5 --warmup 1 --no-fp32-exact --no-f16 --no-cpu-baseline > gpurun_out/bench_c.log 2>&1; echo "bench exit: $?"
grep '^{' gpurun_out/bench_c.log > gpurun_out/bench_c.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_c.json'))
print(d['value'],'fps', d['ms_per_step'],'ms', d.get('phases_ms'))
for r in d.get('rooflines',[]): print('  ', r['kind'], r['bound'][:4], r['frac'], r['achieved'], r['unit'], r['kernel_ms_per_step'],'ms/step')
c=d.get('c4',{})
if 'error' in c: print(c)
for k,v in c.get('stages',{}).items(): print(k, {a:b for a,b in v.items() if a not in ('roofline','pipeline','note','solver')}, v.get('roofline',{}).get('frac'))
print(c.get('pipeline_frames_per_s',{}).get('value'), c.get('pipeline_frames_per_s',{}).get('stages_ms'))
PY
