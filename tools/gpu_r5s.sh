#!/bin/bash
# where does the run-to-run difference under GPU sharing come from?  three concurrent processes x 40 passes per configuration
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
run3() {  # label, dir, env...
  label=$1; dir=$2; shift; shift
  for i in 1 2 3; do (cd $dir && env "$@" timeout 500 python tools/determinism_check.py --passes 40 > $R/gpurun_out/r5s_${label}_$i.txt 2>&1 &) ; done
  sleep 3
  while pgrep -f determinism_check.py > /dev/null; do sleep 2; done
  echo "== $label"; for i in 1 2 3; do grep -E "differ from" $R/gpurun_out/r5s_${label}_$i.txt | cut -c1-200; grep -E "^pass " $R/gpurun_out/r5s_${label}_$i.txt | head -2 | cut -c1-160; done
}
run3 round4_tree $R/_r4 A=1
run3 old_k_order $R FGT_HIP_LIB=$R/fgt_amd/lib/libfgt_hip_oldk.so
run3 static_tiles $R FGT_AUTOTUNE=0
run3 wide_off $R FGT_TAPS_WIDE=0
