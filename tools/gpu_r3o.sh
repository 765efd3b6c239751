#!/bin/bash
# round 3, visit O: bf16x3 attention with all K fragments in flight before the first QK^T MFMA (experiment build) — same-box A/B
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
summ() { python - "$1" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print(d["value"],"fps", d["ms_per_step"],"ms", "checksum", d.get("output_checksum"))
for r in d["rooflines"][:3]: print("  ", r["kind"], r["frac"], r["kernel_ms_per_step"])
PY
}
for rep in 1 2; do
  echo "== default"; timeout 600 python bench.py --steps 5 --warmup 1 --no-c4 --no-f16 --no-fp32-exact --no-cpu-baseline > gpurun_out/bench_o_a$rep.log 2>&1; summ gpurun_out/bench_o_a$rep.log
  echo "== kbatch"; FGT_HIP_LIB=$PWD/fgt_amd/lib/libfgt_hip_kbatch.so timeout 600 python bench.py --steps 5 --warmup 1 --no-c4 --no-f16 --no-fp32-exact --no-cpu-baseline > gpurun_out/bench_o_b$rep.log 2>&1; summ gpurun_out/bench_o_b$rep.log
done
FGT_HIP_LIB=$PWD/fgt_amd/lib/libfgt_hip_kbatch.so timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "attention or attn" -p no:cacheprovider 2>&1 | tail -2
