#!/bin/bash
# round 3, visit K: correlation lookup through LDS windows + split output — flow parity tests, RAFT breakdown, C4 stage times
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_flow_gpu.py -q -x -p no:cacheprovider > gpurun_out/pytest_k.log 2>&1
echo "pytest exit: $?"; grep -E "passed|failed|error" gpurun_out/pytest_k.log | tail -2; grep -E "^FAILED|^ERROR|Error" gpurun_out/pytest_k.log | head -20
echo "== RAFT breakdown"
timeout 600 python tools/raft_breakdown.py > gpurun_out/raft_breakdown_k.txt 2>&1; cut -c1-200 gpurun_out/raft_breakdown_k.txt | head -20
