#!/bin/bash
# round 3, visit Q: XCD-aware work order of the split-input attention (query blocks of a problem on one XCD): same-box A/B at 432x240x80 and 864x480x160
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
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "attention or attn" -p no:cacheprovider 2>&1 | tail -1
for rep in 1 2; do
  echo "== plain grid"; FGT_ATTN_XCD=0 timeout 600 python bench.py --steps 5 --warmup 1 --no-c4 --no-f16 --no-fp32-exact --no-cpu-baseline > gpurun_out/bench_q_a$rep.log 2>&1; summ gpurun_out/bench_q_a$rep.log
  echo "== XCD order"; timeout 600 python bench.py --steps 5 --warmup 1 --no-c4 --no-f16 --no-fp32-exact --no-cpu-baseline > gpurun_out/bench_q_b$rep.log 2>&1; summ gpurun_out/bench_q_b$rep.log
done
echo "== 864x480x160, plain grid"; FGT_ATTN_XCD=0 timeout 600 python bench.py --frames 160 --height 480 --width 864 --steps 2 --warmup 1 --no-c4 --no-f16 --no-fp32-exact --no-cpu-baseline > gpurun_out/bench_q_c5a.log 2>&1; summ gpurun_out/bench_q_c5a.log
echo "== 864x480x160, XCD order"; timeout 600 python bench.py --frames 160 --height 480 --width 864 --steps 2 --warmup 1 --no-c4 --no-f16 --no-fp32-exact --no-cpu-baseline > gpurun_out/bench_q_c5b.log 2>&1; summ gpurun_out/bench_q_c5b.log
