#!/bin/bash
# round-5 visit I: K walk order of the tap-reusing kernels: (ky, chunk, kx) vs (chunk, ky, kx)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for ko in 0 1; do
  FGT_TAPS_KORDER=$ko timeout 600 python -m pytest tests/test_taps_gpu.py tests/test_foldconv_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r5i_tests_ko$ko.log 2>&1; echo "tests korder=$ko exit $?"; tail -1 gpurun_out/r5i_tests_ko$ko.log
  FGT_TAPS_KORDER=$ko timeout 280 python tools/conv_breakdown.py > gpurun_out/r5i_conv_breakdown_ko$ko.txt 2>&1; head -2 gpurun_out/r5i_conv_breakdown_ko$ko.txt | tail -1
  FGT_TAPS_KORDER=$ko timeout 200 python tools/fold_conv_micro.py > gpurun_out/r5i_fold_micro_ko$ko.txt 2>&1; grep -E "128x128it|256x128it|128x128x8t|Linear" gpurun_out/r5i_fold_micro_ko$ko.txt | cut -c1-260
done
