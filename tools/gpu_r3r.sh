#!/bin/bash
# round 3, visit R: encode chunk size (frames per call of the per-frame stages): 20 vs 40 vs 80, same box
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
summ() { python - "$1" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print(d["value"],"fps", d["ms_per_step"],"ms", "checksum", d.get("output_checksum"), d.get("phases_ms"))
PY
}
for ck in 20 40 80 20 40; do
  echo "== encode chunk $ck"; timeout 600 python bench.py --steps 5 --warmup 1 --encode-chunk $ck --no-c4 --no-f16 --no-fp32-exact --no-cpu-baseline > gpurun_out/bench_r_$ck.log 2>&1; summ gpurun_out/bench_r_$ck.log
done
