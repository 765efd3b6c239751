#!/bin/bash
# round 3, visit G: tap-reusing conv kernel, version 2 (diagnostic library) — parity, per-layer sweep vs the early-release kernel, bench with routing on
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
export FGT_HIP_LIB=$PWD/fgt_amd/lib/libfgt_hip_diag.so
timeout 600 python -m pytest tests/test_taps_gpu.py -q -rA -p no:cacheprovider > gpurun_out/pytest_taps2.log 2>&1
echo "pytest taps exit: $?"; grep -E "passed|failed|error" gpurun_out/pytest_taps2.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/pytest_taps2.log | head -20; grep "\[parity\]" gpurun_out/pytest_taps2.log | cut -c1-230 | head -24
echo "== sweep"
timeout 600 python tools/split_sweep.py --diag --reps 10 --split-only --layers "e20 enc8,e20 enc10,e20 enc6,dec   128,raft,lafc" --tiles "128x128x8ea,128x128x8t,128x128t,128x64t" > gpurun_out/split_sweep_taps2.txt 2>&1
echo "sweep exit: $?"; cut -c1-200 gpurun_out/split_sweep_taps2.txt
echo "== bench, routing on"
FGT_CONV_TAPS=1 timeout 900 python bench.py --steps 5 --warmup 1 --no-fp32-exact --no-f16 > gpurun_out/bench_g.log 2>&1; echo "bench exit: $?"
grep '^{' gpurun_out/bench_g.log > gpurun_out/bench_g.json; tail -3 gpurun_out/bench_g.log | cut -c1-300
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_g.json'))
print(d['value'],'fps', d['ms_per_step'],'ms', 'parity', d.get('parity_vs_cpu_oracle',{}).get('max_abs_diff'))
for r in d.get('rooflines',[])[:3]: print('  ', r['kind'], r['bound'][:4], r['frac'], r['achieved'], r['unit'], r['kernel_ms_per_step'],'ms/step')
c=d.get('c4',{})
if 'error' in c: print(c)
for k,v in c.get('stages',{}).items(): print(k, {a:b for a,b in v.items() if a not in ('roofline','pipeline','note','solver')}, v.get('roofline',{}).get('frac'))
print(c.get('pipeline_frames_per_s',{}).get('value'), c.get('pipeline_frames_per_s',{}).get('stages_ms'))
PY
