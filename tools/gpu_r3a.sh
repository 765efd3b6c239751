#!/bin/bash
# round 3, visit A: wide-kernel parity + sweep, full suite (fp16 attention prefetch variant on by default), bench with c4 + HBM rooflines
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== wide kernel parity"
timeout 600 python -m pytest tests/test_split_gpu.py -q -p no:cacheprovider -k "wide" > gpurun_out/pytest_wide.log 2>&1
echo "pytest wide exit: $?"; tail -4 gpurun_out/pytest_wide.log; grep -E "^FAILED|^ERROR" gpurun_out/pytest_wide.log | head -20
echo "== sweep: narrow planes / interleaved vs wide"
timeout 600 python tools/split_sweep.py --reps 10 --split-only --layers "e20 enc8,e20 enc10,e20 enc6,e20 p2v,dec   128,dec   64,b8 ffn1,b8 qkv,b8 proj,b8 k,v2p" \
   --tiles "128x128x8ea,128x128x8eaw,128x128eaw,256x128x16eaw,256x128eaw,256x64x8eaw,256x256p8,256x256p8w,256x128p8,256x128p8w" > gpurun_out/split_sweep_wide.txt 2>&1
echo "sweep exit: $?"; cut -c1-260 gpurun_out/split_sweep_wide.txt
echo "== full suite"
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=8 -p no:cacheprovider --deselect tests/test_split_gpu.py::test_conv_wide_tiles_bit_equal > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?"; grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/pytest_gpu.log | head -30
grep "\[parity\]" gpurun_out/pytest_gpu.log | grep -E "bench clip|contractive|RAFT 864" | cut -c1-300
echo "== bench"
timeout 900 python bench.py --steps 5 --warmup 1 > gpurun_out/bench.log 2>&1; echo "bench exit: $?"
grep '^{' gpurun_out/bench.log > gpurun_out/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench.json'))
print(d['value'],'fps', d['ms_per_step'],'ms; cpu', d.get('cpu_baseline',{}).get('value'), 'parity', d.get('parity_vs_cpu_oracle',{}).get('max_abs_diff'))
for r in d.get('rooflines',[]): print('  ', r['kind'], r['bound'][:4], r['frac'], r['achieved'], r['unit'], r['kernel_ms_per_step'],'ms/step', r['avg_launch_us'],'us/launch')
for name in ('f16','fp32_exact'):
    f=d.get(name)
    if f: print(name, f['value'],'fps', f['ms_per_step'],'ms')
c=d.get('c4',{})
print(json.dumps({k:v for k,v in c.items() if k!='cpu_baseline'}, indent=0)[:3500])
print(json.dumps(c.get('cpu_baseline'))[:1500])
PY
tail -3 gpurun_out/bench.log | cut -c1-400
