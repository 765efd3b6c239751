#!/bin/bash
# after taking the wide image off the 128x128it tile: stress (3 processes), per-layer check, bench
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3; do (timeout 500 python tools/determinism_check.py --passes 60 $([ $i = 2 ] && echo --graphs) > gpurun_out/r5u_det_$i.txt 2>&1 &); done
sleep 3; while pgrep -f determinism_check.py > /dev/null; do sleep 2; done
for i in 1 2 3; do grep -E "differ from" gpurun_out/r5u_det_$i.txt | cut -c1-160; done
for i in 1 2 3; do (timeout 500 python tools/layer_race_check.py --reps 100 > gpurun_out/r5u_layer_$i.txt 2>&1 &); done
sleep 3; while pgrep -f layer_race_check.py > /dev/null; do sleep 2; done
echo "launches differing from the first (all processes, all layers x tiles):"; cat gpurun_out/r5u_layer_*.txt | grep -c " of 100"; cat gpurun_out/r5u_layer_*.txt | grep " [1-9][0-9]* of 100" | head
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-fp32-exact --no-c4 > gpurun_out/r5u_bench.log 2>&1
python - <<PY
import json; d=json.load(open('gpurun_out/bench_detail.json'))
print(d['value'],'fps', d['ms_per_step'],'ms', [(r['kind'], r['frac'], r['kernel_ms_per_step']) for r in d.get('rooflines',[])[:2]], d['output_checksum'])
PY
FGT_TAPS_WIDE=2 timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-fp32-exact --no-c4 > gpurun_out/r5u_bench_wide128.log 2>&1
python - <<PY
import json; d=json.load(open('gpurun_out/bench_detail.json'))
print('FGT_TAPS_WIDE=2 (wide image also on 128x128it):', d['value'],'fps', d['ms_per_step'],'ms', [(r['kind'], r['frac'], r['kernel_ms_per_step']) for r in d.get('rooflines',[])[:2]], d['output_checksum'])
PY
