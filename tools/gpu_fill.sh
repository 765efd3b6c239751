#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_fill_gpu.py -q -x -rA -p no:cacheprovider > gpurun_out/pytest_fill.log 2>&1
echo "pytest exit: $?"; grep "parity\|passed\|failed\|Error" gpurun_out/pytest_fill.log | tail -20
