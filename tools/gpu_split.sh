#!/bin/bash
# GPU visit for the pre-split / LDS-DMA conv path: parity tests, then the old-vs-new sweep with both schedules.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_split_gpu.py -q -x -p no:cacheprovider > gpurun_out/pytest_split.log 2>&1
echo "pytest exit: $?"; tail -15 gpurun_out/pytest_split.log
FGT_CONV_PIPE=1 timeout 600 python tools/split_sweep.py > gpurun_out/split_sweep_pin1.log 2>&1; cat gpurun_out/split_sweep_pin1.log | cut -c1-220
FGT_CONV_PIPE=0 timeout 600 python tools/split_sweep.py --layers enc8,ffn1,ffn2,qkv > gpurun_out/split_sweep_pin0.log 2>&1; cat gpurun_out/split_sweep_pin0.log | cut -c1-220
