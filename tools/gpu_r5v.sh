#!/bin/bash
# which half of the wide 128x128it instance is not reproducible under sharing: the A request schedule (all in tap 0 + pre-barrier reads) or the image?
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
run3() {
  label=$1; shift
  for i in 1 2 3; do (env "$@" timeout 500 python tools/layer_race_check.py --reps 120 > gpurun_out/r5v_${label}_$i.txt 2>&1 &); done
  sleep 3; while pgrep -f layer_race_check.py > /dev/null; do sleep 2; done
  echo "== $label: 128x128it lines with differing launches / all 128x128it lines"; cat gpurun_out/r5v_${label}_*.txt | grep "128x128it" | grep -c " [1-9][0-9]* of"; cat gpurun_out/r5v_${label}_*.txt | grep -c "128x128it"
  echo "   other tiles with differing launches:"; cat gpurun_out/r5v_${label}_*.txt | grep -v "128x128it" | grep -c " [1-9][0-9]* of"
}
run3 wide128_as_built FGT_TAPS_WIDE=2
run3 wide128_spread_requests FGT_TAPS_WIDE=2 FGT_HIP_LIB=$R/fgt_amd/lib/libfgt_hip_ae0.so
run3 narrow_with_early_requests FGT_TAPS_WIDE=0 FGT_HIP_LIB=$R/fgt_amd/lib/libfgt_hip_ae1.so
