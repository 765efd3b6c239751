#!/bin/bash
# round 6: is instruction fetch a cost of the 50 K-instruction conv kernels?  Counter list first, then one pass per counter group over one step.
R=$(pwd); export TMPDIR=/tmp
cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|IFETCH|WAIT_INST|INST_LEVEL|SQ_BUSY_CY|SQ_WAVE_CYCLES|SQ_INSTS_VALU\b|SQ_INSTS_SALU\b|SQ_ACTIVE_INST|INST_CYCLES" | cut -c1-160 | sort -u | head -60 > "$R/gpurun_out/r06_counters_avail.txt"
cd "$R"
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_SALU"; do
  tag=$(echo $grp | tr ' ' '_')
  (cd /tmp && FGT_TUNING_FILE="$R/gpurun_out/tuning.json" timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$R/gpurun_out/pmci_$tag" -o pmc -- python "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-prof --no-fp32-exact --no-f16 --no-c4 > "$R/gpurun_out/pmci_$tag.log" 2>&1)
  echo "pmc $grp exit $?"
  python tools/pmc_any.py gpurun_out/pmci_$tag --top 8 > gpurun_out/r06_icache_$tag.txt 2>&1
  find gpurun_out/pmci_$tag -name "*.csv" -size +8M -delete
done
cat gpurun_out/r06_counters_avail.txt | head -40
cat gpurun_out/r06_icache_*.txt | cut -c1-260
