#!/bin/bash
# round 3, visit B: on-chip solvers (tests + c4 bench), non-temporal epilogue stores A/B, RAFT pair-batch sweep
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== on-chip solver tests"
timeout 900 python -m pytest tests/test_fill_gpu.py tests/test_blend_pinned.py tests/test_c1_plumbing.py -m gpu -q -rA -p no:cacheprovider > gpurun_out/pytest_solvers.log 2>&1
echo "pytest exit: $?"; grep -E "passed|failed|error" gpurun_out/pytest_solvers.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/pytest_solvers.log | head; grep "\[parity\]" gpurun_out/pytest_solvers.log | grep -E "on-chip" | cut -c1-250
echo "== bench (headline only + c4)"
timeout 900 python bench.py --steps 5 --warmup 1 --no-fp32-exact --no-f16 > gpurun_out/bench_b.log 2>&1; echo "bench exit: $?"
grep '^{' gpurun_out/bench_b.log > gpurun_out/bench_b.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_b.json'))
print(d['value'],'fps', d['ms_per_step'],'ms')
for r in d.get('rooflines',[]): print('  ', r['kind'], r['bound'][:4], r['frac'], r['achieved'], r['unit'], r['kernel_ms_per_step'],'ms/step')
c=d.get('c4',{})
if 'error' in c: print(c)
for k,v in c.get('stages',{}).items(): print(k, {a:b for a,b in v.items() if a not in ('roofline','pipeline','note')}, v.get('roofline',{}).get('frac'))
print(c.get('pipeline_frames_per_s'))
PY
echo "== non-temporal epilogue stores A/B (same process order: NT=0 then NT=1)"
for nt in 0 1; do
  FGT_CONV_NT=$nt timeout 300 python tools/split_sweep.py --reps 10 --split-only --layers "e20 enc8,e20 enc10,dec   128,b8 ffn1,b8 qkv,b8 proj,b8 k,b8 ffn2,v2p" --tiles "128x128x8ea,128x128x8" > gpurun_out/split_sweep_nt$nt.txt 2>&1
  echo "NT=$nt"; cut -c1-120 gpurun_out/split_sweep_nt$nt.txt
done
echo "== RAFT pair batch"
timeout 600 python tools/raft_batch.py > gpurun_out/raft_batch.txt 2>&1; cat gpurun_out/raft_batch.txt | cut -c1-200
