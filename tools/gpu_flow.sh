#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_flow_gpu.py -m gpu -q -rA -p no:cacheprovider > gpurun_out/pytest_flow.log 2>&1
grep -E "parity|passed|failed|Error" gpurun_out/pytest_flow.log | cut -c1-170 | tail -20
timeout 900 python tools/flow_bench.py 2>&1 | grep case | cut -c1-220
