#!/bin/bash
# same-box A/B: FGT_UP4=1 vs 0 at the driver's settings + LAFC + pipeline
for r in 1 2; do for v in 1 0; do
  FGT_UP4=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-exact --no-f16 > gpurun_out/ab_up4_$v.log 2>&1
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_detail.json"))
c4=d.get("c4",{}); st=c4.get("stages",{})
print("FGT_UP4=$v", d["value"],"fps", d["ms_per_step"],"ms | conv", d["roofline"]["kernel_ms_per_step"], "frac", d["roofline"]["frac"], "checksum", d.get("output_checksum"), "| lafc", st.get("lafc",{}).get("ms_per_flow"), "raft", st.get("raft_864x480",{}).get("ms_per_pair"), "pipeline", (c4.get("pipeline_frames_per_s") or {}).get("value"), "parity", (d.get("parity_vs_cpu_oracle") or {}).get("max_abs_diff"))
PY
done; done
