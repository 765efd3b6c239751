#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3; do (timeout 800 python tools/determinism_flow_check.py --passes 8 > gpurun_out/r5x_flow_$i.txt 2>&1 &); done
sleep 3; while pgrep -f determinism_flow_check.py > /dev/null; do sleep 2; done
for i in 1 2 3; do tail -3 gpurun_out/r5x_flow_$i.txt | cut -c1-200; done
