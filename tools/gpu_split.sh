#!/bin/bash
# GPU visit for the pre-split conv path: parity tests, then the old-vs-new sweep.   usage: gpu_split.sh "<layer filter>" [pipe]
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
export FGT_CONV_PIPE=${2:-1}
timeout 900 python -m pytest tests/test_split_gpu.py -q -x -p no:cacheprovider > gpurun_out/pytest_split.log 2>&1
echo "pytest exit: $? (FGT_CONV_PIPE=$FGT_CONV_PIPE)"; tail -5 gpurun_out/pytest_split.log
timeout 600 python tools/split_sweep.py --layers "${1:-}" > gpurun_out/split_sweep_pipe$FGT_CONV_PIPE.log 2>&1; cat gpurun_out/split_sweep_pipe$FGT_CONV_PIPE.log | cut -c1-230
