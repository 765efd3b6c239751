#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== graph tests"
timeout 900 python -m pytest tests/test_graph_gpu.py -m gpu -q -rA -p no:cacheprovider 2>&1 | tail -15 | cut -c1-200
echo "== bench eager vs graphs"
for g in "" "--graphs"; do
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $g > gpurun_out/bench_g.log 2>&1
grep '^{' gpurun_out/bench_g.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['hip_graphs'], d['value'],'fps', d['ms_per_step'],'ms host', d['host_enqueue_ms_per_step'], d['roofline']['achieved'], d['output_checksum'])" || tail -5 gpurun_out/bench_g.log
done
echo "== flow bench eager vs graphs"
timeout 600 python tools/flow_bench.py 2>&1 | grep case | cut -c1-200
FGT_GRAPHS=1 timeout 600 python tools/flow_bench.py 2>&1 | grep case | cut -c1-200
