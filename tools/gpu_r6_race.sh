#!/bin/bash
# Round 6, visit 2: the wide image + early schedule back on after the retire_pre_reads fix — 3-process stress (whole step and per layer x tile),
# then the same-box A/B of the wide image at the driver's settings.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_run2
echo "== 3 processes x 150 passes of the FGT step (default = wide image, early schedule)" | tee ${O}_stress.txt
for i in 1 2 3; do (timeout 900 python tools/determinism_check.py --passes 150 > ${O}_det_p$i.txt 2>&1) & done; wait
tail -n 2 ${O}_det_p*.txt | tee -a ${O}_stress.txt
echo "== 3 processes x layer_race_check --reps 150" | tee -a ${O}_stress.txt
for i in 1 2 3; do (timeout 900 python tools/layer_race_check.py --reps 150 > ${O}_race_p$i.txt 2>&1) & done; wait
grep -h "launches differ" ${O}_race_p*.txt | grep -v ": 0 of" | head -20 | tee -a ${O}_stress.txt
echo "lines with 0 differing: $(grep -h 'launches differ' ${O}_race_p*.txt | grep -c ': 0 of')  with > 0: $(grep -h 'launches differ' ${O}_race_p*.txt | grep -vc ': 0 of')" | tee -a ${O}_stress.txt
echo "== A/B wide image at --steps 20 --warmup 5"
for w in 1 0 1 0; do
  FGT_TAPS_WIDE=$w timeout 600 python bench.py --steps 20 --warmup 5 --no-c4 --no-fp32-exact --no-cpu-baseline > ${O}_bench_wide$w.log 2>&1
  python - <<P | tee -a ${O}_ab.txt
import json; d=json.load(open('gpurun_out/bench_detail.json')); r=d['roofline']
print('FGT_TAPS_WIDE=$w', d['value'], 'fps', d['ms_per_step'], 'ms; conv', r['kernel_ms_per_step'], 'ms frac', r['frac'], 'clock', r['sustained']['clock_ghz'])
P
done
