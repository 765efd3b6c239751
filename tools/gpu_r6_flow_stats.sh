#!/bin/bash
# round 6: rocprofv3 kernel stats of the flow stages on the final tree: RAFT over the 80-frame clip at 864x480 (158 pairs x 20 iterations) and LAFC's completion
# of one 80-frame direction (tools/raft_clip.py, tools/lafc_batch.py), tiles tuned by a first untraced run
R=$(pwd); export TMPDIR=/tmp
export FGT_TUNING_FILE="$R/gpurun_out/tuning_flow.json"
FGT_TUNING_SAVE=1 python tools/raft_clip.py --reps 1 > gpurun_out/flow_stats_prep.log 2>&1
FGT_TUNING_SAVE=1 python tools/lafc_batch.py >> gpurun_out/flow_stats_prep.log 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_raft" -o raft -- python "$R/tools/raft_clip.py" --reps 1 > "$R/gpurun_out/prof_raft.log" 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_lafc" -o lafc -- python "$R/tools/lafc_batch.py" > "$R/gpurun_out/prof_lafc.log" 2>&1)
for s in raft lafc; do f=$(find gpurun_out/prof_$s -name "*kernel_stats.csv" | head -1); echo "== $s: $f"; head -14 "$f" | cut -c1-200; cp "$f" gpurun_out/r06_flow_${s}_kernel_stats.csv; find gpurun_out/prof_$s -name "*kernel_trace.csv" -size +8M -delete; done
