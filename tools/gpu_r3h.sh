#!/bin/bash
# round 3, visit H: tap-reusing conv kernel with register-fed weights (diagnostic library) — parity, per-layer sweep
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
export FGT_HIP_LIB=$PWD/fgt_amd/lib/libfgt_hip_diag.so
timeout 600 python -m pytest tests/test_taps_gpu.py -q -x -p no:cacheprovider > gpurun_out/pytest_taps3.log 2>&1
echo "pytest taps exit: $?"; grep -E "passed|failed|error" gpurun_out/pytest_taps3.log | tail -2; grep -E "^FAILED|^ERROR|Error|assert" gpurun_out/pytest_taps3.log | head -20
echo "== sweep"
timeout 900 python tools/split_sweep.py --diag --reps 10 --split-only --layers "e20 enc8,e20 enc10,e20 enc6,dec   128,dec   64,raft,lafc" --tiles "128x128x8ea,128x128x8t,128x64t,64x64t,128x128x8r,128x128r,128x64r,64x64r" > gpurun_out/split_sweep_taps3.txt 2>&1
echo "sweep exit: $?"; cut -c1-260 gpurun_out/split_sweep_taps3.txt
