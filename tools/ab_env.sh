#!/bin/bash
# Same-box A/B of an environment switch over the FGT stage of the bench: tools/ab_env.sh VAR val_a val_b [kinds...]   (run on the GPU box)
# (round 6: at the driver's settings by default, --steps 20 --warmup 5: AB_STEPS / AB_WARMUP override)
# Prints fps, ms/step and the per-kernel ms of the named roofline kinds for both values, two rounds each (box drift shows as the spread).
VAR=$1; A=$2; B=$3; shift 3
for round in 1 2; do for v in $A $B; do
  env $VAR=$v FGT_TUNING_FILE=$PWD/gpurun_out/tuning.json python bench.py --steps ${AB_STEPS:-20} --warmup ${AB_WARMUP:-5} --no-cpu-baseline --no-fp32-exact --no-c4 > /tmp/ab.log 2>&1
  python - "$VAR=$v" "$@" <<'P'
import json, sys
d = json.load(open('gpurun_out/bench_detail.json'))
k = {r['kind']: r['kernel_ms_per_step'] for r in d.get('rooflines', [])}
print(sys.argv[1], d['value'], 'fps', d['ms_per_step'], 'ms |', ' '.join(f"{x} {k.get(x)}" for x in sys.argv[2:]), '| checksum', d['output_checksum'])
P
done; done
