#!/bin/bash
# round 6: LDS bank conflicts of the MFMA kernels over one step (rocprofv3 --pmc; tiles pre-seeded by a first untraced run)
R=$(pwd); export TMPDIR=/tmp
export FGT_TUNING_FILE="$R/gpurun_out/tuning_lds.json"
FGT_TUNING_SAVE=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-prof --no-fp32-exact --no-f16 --no-c4 > gpurun_out/lds_prep.log 2>&1
cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -i -E "Counter_Name.*LDS" | sort -u | head -30 > "$R/gpurun_out/r06_lds_counters_avail.txt"; cd "$R"
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES"; do
  tag=$(echo $grp | tr ' ' '_')
  (cd /tmp && timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$R/gpurun_out/pmcl_$tag" -o pmc -- python "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-prof --no-fp32-exact --no-f16 --no-c4 > "$R/gpurun_out/pmcl_$tag.log" 2>&1)
  echo "pmc $grp exit $?"
  python tools/pmc_any.py gpurun_out/pmcl_$tag --top 8 > gpurun_out/r06_lds_$tag.txt 2>&1
  find gpurun_out/pmcl_$tag -name "*.csv" -size +8M -delete
done
cat gpurun_out/r06_lds_counters_avail.txt | cut -c1-100; cat gpurun_out/r06_lds_*.txt | cut -c1-260
