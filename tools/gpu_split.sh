#!/bin/bash
# GPU visit for the pre-split / LDS-DMA conv path: parity tests, then the old-vs-new sweep.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_split_gpu.py -q -x -p no:cacheprovider > gpurun_out/pytest_split.log 2>&1
echo "pytest exit: $?"; tail -5 gpurun_out/pytest_split.log
FGT_CONV_PIPE=1 timeout 600 python tools/split_sweep.py --layers "${1:-}" > gpurun_out/split_sweep_pin1.log 2>&1; cat gpurun_out/split_sweep_pin1.log | cut -c1-220
