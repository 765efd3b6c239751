#!/bin/bash
# round 3, visit M: conv epilogue without store-acknowledgement waits (per-aux-count bodies, aux loads one group ahead): sweep, full GPU suite, bench
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== sweep"
timeout 600 python tools/split_sweep.py --reps 10 --split-only --layers "b8 ffn1,b8 qkv,b8 proj,b8 k,e20 enc8,e20 enc10,dec   128,raft gru" --tiles "128x128x8ea,128x128x8eaw,128x128x8t,128x64t" > gpurun_out/split_sweep_m.txt 2>&1
echo "sweep exit: $?"; cut -c1-200 gpurun_out/split_sweep_m.txt
echo "== bench"
timeout 900 python bench.py --steps 5 --warmup 1 --no-fp32-exact --no-f16 > gpurun_out/bench_m.log 2>&1; echo "bench exit: $?"
grep '^{' gpurun_out/bench_m.log > gpurun_out/bench_m.json; tail -2 gpurun_out/bench_m.log | cut -c1-200
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_m.json'))
print(d['value'],'fps', d['ms_per_step'],'ms', 'parity', d.get('parity_vs_cpu_oracle',{}).get('max_abs_diff'))
for r in d.get('rooflines',[])[:3]: print('  ', r['kind'], r['bound'][:4], r['frac'], r['achieved'], r['unit'], r['kernel_ms_per_step'],'ms/step')
c=d.get('c4',{})
if 'error' in c: print(c)
for k,v in c.get('stages',{}).items(): print(k, {a:b for a,b in v.items() if a not in ('roofline','pipeline','note','solver')}, v.get('roofline',{}).get('frac'))
print(c.get('pipeline_frames_per_s',{}).get('value'), c.get('pipeline_frames_per_s',{}).get('stages_ms'))
PY
echo "== full GPU suite"
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_m.log 2>&1
echo "pytest exit: $?"; grep -E "passed|failed|error" gpurun_out/pytest_m.log | tail -2; grep -E "^FAILED|^ERROR|Error" gpurun_out/pytest_m.log | head -20
