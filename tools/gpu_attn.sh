#!/bin/bash
# attention kernel visit: parity of the attention tests, micro-benchmarks (4- and 8-wave blocks), multi-rank rehearsal
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== attention + dist tests"
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_dist_gpu.py tests/test_fgt_gpu.py -m gpu -q -rA -k "attention or dist or ranks or bf16x3" -p no:cacheprovider > gpurun_out/pytest_attn.log 2>&1
grep -E "parity.*(attn|clip|bf16x3)|passed|failed|Error" gpurun_out/pytest_attn.log | cut -c1-160 | tail -24
echo "== attention micro"
for nw8 in 0 1; do FGT_ATTN_NW8=$nw8 python tools/attn_micro.py --precision bf16x3; done
python tools/attn_micro.py --precision bf16x3 --t 13; python tools/attn_micro.py --precision bf16x3 --t 18
python tools/attn_micro.py --precision bf16x3 --spatial
for nw8 in 0 1; do
echo "== bench NW8=$nw8"
FGT_ATTN_NW8=$nw8 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_nw8_$nw8.log 2>&1
grep '^{' gpurun_out/bench_nw8_$nw8.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'],'fps', d['ms_per_step'],'ms host', d['host_enqueue_ms_per_step'], d['roofline']['achieved'], d['roofline']['traffic'])"
done
