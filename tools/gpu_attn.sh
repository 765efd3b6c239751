#!/bin/bash
# attention kernel profiling visit + multi-rank rehearsal
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== dist rehearsal test"
timeout 900 python -m pytest tests/test_dist_gpu.py -m gpu -q -rA -p no:cacheprovider 2>&1 | grep -E "parity|passed|failed|Error" | tail -8
echo "== torchrun 2 ranks on one GPU (gloo) vs 1 rank"
FGT_BENCH_SHARE_GPU=1 FGT_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 1 --warmup 0 --no-cpu-baseline --no-prof > gpurun_out/bench_2rank_gloo.log 2>&1; echo "exit $?"
grep '^{' gpurun_out/bench_2rank_gloo.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['value'], d['output_checksum'], d['output_sane'])" || tail -5 gpurun_out/bench_2rank_gloo.log
timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-prof > gpurun_out/bench_1rank.log 2>&1
grep '^{' gpurun_out/bench_1rank.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['value'], d['output_checksum'], d['output_sane'])"
echo "== attention micro"
for p in fp32 bf16x3; do python tools/attn_micro.py --precision $p; python tools/attn_micro.py --precision $p --spatial; done
(cd /tmp
 for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_INSTS_SALU" "FETCH_SIZE"; do
   tag=$(echo $set | cut -d' ' -f1)
   timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/apmc_$tag" -o pmc -- python "$GRAFT_REPO_ROOT/tools/attn_micro.py" --reps 5 > "$GRAFT_REPO_ROOT/gpurun_out/apmc_$tag.log" 2>&1
 done)
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/apmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "attn" in r.get("Kernel_Name", ""):
            a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    print({k: round(v[1] / max(v[0], 1), 1) for k, v in agg.items()})
PY
