#!/bin/bash
# final visit of round 5 (committed tree): 3-process stress, then the full check (tests incl. the sharing tests, smoke, bench, PMC, stats)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3; do (timeout 900 python tools/determinism_check.py --passes 120 > gpurun_out/r5fin_det_$i.txt 2>&1 &); done
sleep 3; while pgrep -f determinism_check.py > /dev/null; do sleep 2; done
for i in 1 2 3; do grep -E "differ from" gpurun_out/r5fin_det_$i.txt | cut -c1-140; grep -E "^pass " gpurun_out/r5fin_det_$i.txt | head -3 | cut -c1-140; done
bash tools/gpu_check.sh test smoke bench pmc stats
