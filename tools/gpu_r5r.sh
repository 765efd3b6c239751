#!/bin/bash
# rare run-to-run differences under GPU sharing: three concurrent processes, 60 passes each, in four configurations
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run3() {  # label, env...
  label=$1; shift
  for i in 1 2 3; do (env "$@" timeout 500 python tools/determinism_check.py --passes 60 $([ $i = 2 ] && echo --graphs) > gpurun_out/r5r_${label}_$i.txt 2>&1 &) ; done
  wait; sleep 1
  while pgrep -f determinism_check.py > /dev/null; do sleep 2; done
  echo "== $label"; for i in 1 2 3; do grep -E "differ|pass " gpurun_out/r5r_${label}_$i.txt | tail -4 | cut -c1-200; done
}
run3 default A=1
run3 notaps FGT_CONV_TAPS=0
run3 nofold FGT_FOLD_CONV=0
