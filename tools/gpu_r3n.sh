#!/bin/bash
# round 3, visit N: LAFC breakdown, GEMM sweep with the new epilogue, taps tests
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_taps_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -2
echo "== LAFC breakdown"
timeout 600 python tools/lafc_breakdown.py > gpurun_out/lafc_breakdown.txt 2>&1; cut -c1-200 gpurun_out/lafc_breakdown.txt | head -45
echo "== sweep"
timeout 600 python tools/split_sweep.py --reps 10 --split-only --layers "b8 ffn1,b8 qkv,b8 proj,b8 k,b8 ffn2,e20 enc8" --tiles "128x128x8ea,128x128x8eaw,128x128x8,128x128,128x64ea" > gpurun_out/split_sweep_n.txt 2>&1
echo "sweep exit: $?"; cut -c1-200 gpurun_out/split_sweep_n.txt
