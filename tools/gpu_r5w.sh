#!/bin/bash
# wide image on the 128x128it tile with the spread request schedule: stress (layers and the whole step, 3 processes), then the bench
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3; do (timeout 500 python tools/layer_race_check.py --reps 150 > gpurun_out/r5w_layer_$i.txt 2>&1 &); done
sleep 3; while pgrep -f layer_race_check.py > /dev/null; do sleep 2; done
echo "layer x tile lines: $(cat gpurun_out/r5w_layer_*.txt | grep -c ' of 150'), with differing launches: $(cat gpurun_out/r5w_layer_*.txt | grep -c ' [1-9][0-9]* of 150')"
for i in 1 2 3; do (timeout 500 python tools/determinism_check.py --passes 80 $([ $i = 2 ] && echo --graphs) > gpurun_out/r5w_det_$i.txt 2>&1 &); done
sleep 3; while pgrep -f determinism_check.py > /dev/null; do sleep 2; done
for i in 1 2 3; do grep -E "differ from" gpurun_out/r5w_det_$i.txt | cut -c1-160; done
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-fp32-exact --no-c4 > gpurun_out/r5w_bench.log 2>&1
python - <<PY
import json; d=json.load(open('gpurun_out/bench_detail.json'))
print(d['value'],'fps', d['ms_per_step'],'ms', [(r['kind'], r['frac'], r['kernel_ms_per_step']) for r in d.get('rooflines',[])[:2]], d['output_checksum'])
PY
FGT_TAPS_WIDE=0 timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-fp32-exact --no-c4 > gpurun_out/r5w_bench_narrow.log 2>&1
python - <<PY
import json; d=json.load(open('gpurun_out/bench_detail.json'))
print('FGT_TAPS_WIDE=0:', d['value'],'fps', d['ms_per_step'],'ms', [(r['kind'], r['frac'], r['kernel_ms_per_step']) for r in d.get('rooflines',[])[:2]], d['output_checksum'])
PY
