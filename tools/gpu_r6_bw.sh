#!/bin/bash
# Round 6: the bandwidth kernels after the XCD-contiguous work order (warp, fb check, Cout <= 4 conv) and the 32-channel chunks of the small conv:
# parity tests, event timings, rocprofv3 counter traffic (FETCH_SIZE / WRITE_SIZE in separate passes) of tools/hbm_micro.py and of a small-conv micro.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp; R="$PWD"; O=gpurun_out/r06_run7
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_flow_gpu.py tests/test_fgt_gpu.py tests/test_clip_gpu.py -m gpu -x -q -p no:cacheprovider > ${O}_pytest.log 2>&1; tail -3 ${O}_pytest.log
python tools/hbm_micro.py ${O}_hbm_micro_alg.json 2>&1 | tail -1
for c in 1 0; do FGT_CONV_SMALL_C32=$c python tools/conv_small_micro.py 2>&1 | tail -3; done
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/r06_pmcm_$c" -o pmc -- python "$R/tools/hbm_micro.py" /tmp/alg_$c.json > "$R/${O}_pmcm_$c.log" 2>&1); echo "pmc hbm_micro $c exit $?"
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/r06_pmcs_$c" -o pmc -- python "$R/tools/conv_small_micro.py" > "$R/${O}_pmcs_$c.log" 2>&1); echo "pmc conv_small $c exit $?"
done
python tools/pmc_sum.py gpurun_out/r06_pmcm_FETCH_SIZE gpurun_out/r06_pmcm_WRITE_SIZE warp_ corr_lookup fb_kernel | tee ${O}_traffic_hbm_micro.txt
python tools/pmc_sum.py gpurun_out/r06_pmcs_FETCH_SIZE gpurun_out/r06_pmcs_WRITE_SIZE conv3x3_tiled | tee ${O}_traffic_conv_small.txt
