#!/bin/bash
# round 3, visit E: interleaved split tensors end to end (parity + bench), RAFT breakdown on split chains
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_split_gpu.py tests/test_fgt_gpu.py tests/test_clip_gpu.py -m gpu -q -rA -p no:cacheprovider -k "interleaved or fgt or cliprunner_cache" --deselect tests/test_split_gpu.py::test_conv_interleaved_inputs_bit_equal > gpurun_out/pytest_e.log 2>&1
echo "pytest exit: $?"; grep -E "passed|failed|error" gpurun_out/pytest_e.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/pytest_e.log | head
echo "== bench SPLIT_IL=1 (default)"
timeout 900 python bench.py --steps 5 --warmup 1 --no-fp32-exact --no-f16 --no-cpu-baseline --no-c4 > gpurun_out/bench_e1.log 2>&1; echo "bench exit: $?"
grep '^{' gpurun_out/bench_e1.log > gpurun_out/bench_e1.json
echo "== bench SPLIT_IL=0"
FGT_SPLIT_IL=0 timeout 900 python bench.py --steps 5 --warmup 1 --no-fp32-exact --no-f16 --no-cpu-baseline --no-c4 > gpurun_out/bench_e0.log 2>&1; echo "bench exit: $?"
grep '^{' gpurun_out/bench_e0.log > gpurun_out/bench_e0.json
python - <<'PY'
import json
for n in ('e1','e0'):
    d=json.load(open(f'gpurun_out/bench_{n}.json'))
    print(n, d['value'],'fps', d['ms_per_step'],'ms', d['output_checksum'])
    for r in d.get('rooflines',[])[:3]: print('  ', r['kind'], r['bound'][:4], r['frac'], r['achieved'], r['unit'], r['kernel_ms_per_step'],'ms/step')
PY
python - <<'PY'
import json, collections
t=json.load(open('gpurun_out/tuning.json'))
c=collections.Counter(t.values()); print('tiles picked (last bench):', sorted(c.items()))
PY
echo "== RAFT breakdown"
timeout 600 python tools/raft_breakdown.py > gpurun_out/raft_breakdown.txt 2>&1; cut -c1-200 gpurun_out/raft_breakdown.txt | head -28
