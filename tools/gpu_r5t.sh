#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3; do (timeout 500 python tools/layer_race_check.py > gpurun_out/r5t_layer_race_$i.txt 2>&1 &); done
sleep 3; while pgrep -f layer_race_check.py > /dev/null; do sleep 2; done
cat gpurun_out/r5t_layer_race_1.txt | grep -v amdgpu.ids; grep -h " [1-9][0-9]* of" gpurun_out/r5t_layer_race_2.txt gpurun_out/r5t_layer_race_3.txt | head -20
