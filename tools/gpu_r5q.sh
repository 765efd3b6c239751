#!/bin/bash
# 2-rank rehearsal repeated: are the sharded composites of the two exchanges reproducible (graphs on / off)?
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for g in on on off; do
  FGT_BENCH_SHARE_GPU=1 FGT_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29547 \
    bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-fp32-exact --no-c4 --graphs $g > gpurun_out/r5q_rehearsal_$g.log 2>&1
  grep '^{' gpurun_out/r5q_rehearsal_$g.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('graphs $g', d['output_checksum'], {k:(v.get('output_checksum'), v.get('value')) for k,v in d['strong_scaling_modes'].items()}, d.get('weak_scaling_clip_per_rank',{}).get('output_checksum'))"
done
