#!/bin/bash
# round 3, visit I: tap-reusing conv kernel — 256x128 (16 wavefronts) and 128x64 (8 wavefronts) tiles
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
export FGT_HIP_LIB=$PWD/fgt_amd/lib/libfgt_hip_diag.so
timeout 900 python tools/split_sweep.py --diag --reps 10 --split-only --layers "e20 enc8,e20 enc10,e20 enc6,dec   128,dec   64,raft,lafc" --tiles "128x128x8ea,128x128x8t,128x64t,256x128x16t,128x64x8t,128x128t" > gpurun_out/split_sweep_taps4.txt 2>&1
echo "sweep exit: $?"; cut -c1-260 gpurun_out/split_sweep_taps4.txt
