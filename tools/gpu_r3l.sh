#!/bin/bash
# round 3, visit L: tap-reusing kernel in transposed mode (k x 1 convolutions), correlation lookup with 10-row windows
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_taps_gpu.py tests/test_flow_gpu.py -q -x -p no:cacheprovider > gpurun_out/pytest_l.log 2>&1
echo "pytest exit: $?"; grep -E "passed|failed|error" gpurun_out/pytest_l.log | tail -2; grep -E "^FAILED|^ERROR|Error" gpurun_out/pytest_l.log | head -20
echo "== sweep"
timeout 600 python tools/split_sweep.py --reps 10 --split-only --layers "raft gru" --tiles "128x128x8ea,128x64ea,128x128x8t,128x64t,128x64x8t,64x64t" > gpurun_out/split_sweep_l.txt 2>&1
echo "sweep exit: $?"; cut -c1-220 gpurun_out/split_sweep_l.txt
echo "== RAFT breakdown"
timeout 600 python tools/raft_breakdown.py > gpurun_out/raft_breakdown_l.txt 2>&1; cut -c1-200 gpurun_out/raft_breakdown_l.txt | head -16
