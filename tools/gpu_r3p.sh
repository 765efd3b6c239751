#!/bin/bash
# round 3, visit P: bandwidth kernels (Cout <= 4 conv with register prefetch, LayerNorm 4 rows per wavefront, dw_pool weights through LDS): same-box A/B + tests
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
summ() { python - "$1" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print(d["value"],"fps", d["ms_per_step"],"ms", "checksum", d.get("output_checksum"))
for r in d["rooflines"]: print("  ", r["kind"], r["frac"], r["kernel_ms_per_step"])
PY
}
for rep in 1 2; do
  echo "== base"; FGT_HIP_LIB=$PWD/fgt_amd/lib/libfgt_hip_base.so timeout 600 python bench.py --steps 5 --warmup 1 --no-c4 --no-f16 --no-fp32-exact --no-cpu-baseline > gpurun_out/bench_p_a$rep.log 2>&1; summ gpurun_out/bench_p_a$rep.log
  echo "== new"; timeout 600 python bench.py --steps 5 --warmup 1 --no-c4 --no-f16 --no-fp32-exact --no-cpu-baseline > gpurun_out/bench_p_b$rep.log 2>&1; summ gpurun_out/bench_p_b$rep.log
done
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -2
