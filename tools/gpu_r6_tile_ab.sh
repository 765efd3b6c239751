#!/bin/bash
for r in 1 2; do
for v in "" "256x256it,256x128it,128x128it" "256x256it,256x128it" "128x128it,128x64x8t,128x64t"; do
  rm -f /tmp/tune_$$.json
  FGT_TAPS_ONLY="$v" FGT_TUNING_FILE=/tmp/tune_$$.json python tools/sustained_tile_ab.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-exact --no-f16 --no-c4 > gpurun_out/tile_ab.log 2>&1
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_detail.json"))
print("taps only [$v]", d["value"],"fps", d["ms_per_step"],"ms | conv", d["roofline"]["kernel_ms_per_step"], "clock", d["step_profile"]["mfma_probe_clock_ghz"], "checksum", d.get("output_checksum"))
PY
done; done
