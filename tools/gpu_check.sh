#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench (headline + fp32_exact + cpu baseline in one line), PMC traffic of the MFMA
# kernels with pre-seeded tiles, rocprofv3 kernel stats, attention micro-benchmarks, 2-rank rehearsal.  Everything lands in
# gpurun_out/.   usage (from the repo root on the GPU box):  bash tools/gpu_check.sh [stages]
#   stages: any of  test smoke bench pmc pmc16 stats stats16 attn rehearse c5 flow   (default: test smoke bench pmc stats attn rehearse)
set -u
STAGES="${*:-test smoke bench pmc stats attn rehearse}"
has() { [[ " $STAGES " == *" $1 "* ]]; }
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
if has test; then
  echo "== pytest -m gpu"
  timeout 1500 python -m pytest tests -m gpu -q -rA --durations=8 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit: $?" | tee -a gpurun_out/pytest_gpu.log
  grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -3
  grep -E "^FAILED|^ERROR" gpurun_out/pytest_gpu.log | head -20
  grep "\[parity\]" gpurun_out/pytest_gpu.log | grep -E "long|ClipRunner|C2|RAFT 864|clip " | cut -c1-260
fi
if has smoke; then
  echo "== smoke"
  timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?" | tee -a gpurun_out/smoke.log
  tail -2 gpurun_out/smoke.log
fi
if has bench; then
  echo "== bench (default: bf16x3 headline + fp32_exact + cpu baseline)"
  timeout 900 python bench.py --steps 5 --warmup 1 > gpurun_out/bench.log 2>&1; echo "bench exit: $?" | tee -a gpurun_out/bench.log
  cp gpurun_out/bench_detail.json gpurun_out/bench_detail_default.json 2>/dev/null
  tail -1 gpurun_out/bench.log | wc -c
  python -c "
import sys,json; d=json.load(open('gpurun_out/bench_detail.json'))
print(d['value'],'fps', d['ms_per_step'],'ms; cpu', d.get('cpu_baseline',{}).get('value'), 'parity', d.get('parity_vs_cpu_oracle',{}).get('max_abs_diff'))
c4=d.get('c4',{})
print('c4', {k:{kk:vv for kk,vv in v.items() if kk.startswith('ms_per')} for k,v in c4.get('stages',{}).items()}, (c4.get('pipeline_frames_per_s') or {}).get('value'), c4.get('error'))
for r in d.get('rooflines',[]): print('  ', r['kind'], r['frac'], r.get('algorithmic_tflops', r.get('achieved')),r['unit'], r['kernel_ms_per_step'],'ms/step', r['avg_launch_us'],'us/launch')
for name in ('f16', 'fp32_exact'):
    f=d.get(name)
    if f:
        print(name, f['value'],'fps', f['ms_per_step'],'ms', 'parity', (f.get('parity_vs_cpu_oracle') or {}).get('max_abs_diff'), f.get('composite_vs_headline'))
        for r in f.get('rooflines',[]): print('  ', r['kind'], r['frac'], r.get('algorithmic_tflops', r.get('achieved')),r['unit'], r['kernel_ms_per_step'],'ms/step')
" || tail -5 gpurun_out/bench.log
fi
if has pmc && grep -q dirty .git_head 2>/dev/null; then
  echo "== PMC stage REFUSED: .git_head = $(cat .git_head) — profiles/kernel_traffic.json must come from a committed tree (commit, then tools/gpu.sh)"
elif has pmc; then
  echo "== PMC HBM traffic of the MFMA kernels over bench.py (tiles pre-seeded: only clip-pass launches in the trace; tree $(cat .git_head 2>/dev/null))"
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && FGT_TUNING_FILE="$R/gpurun_out/tuning.json" timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmcb_$c" -o pmc -- python "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-prof --no-fp32-exact --no-f16 --no-c4 > "$R/gpurun_out/pmcb_$c.log" 2>&1)
    echo "pmc $c exit $?"
  done
  python tools/pmc_traffic.py gpurun_out/pmcb_FETCH_SIZE gpurun_out/pmcb_WRITE_SIZE bf16x3 gpurun_out/kernel_traffic.json
  echo "== PMC HBM traffic of the flow stages' bandwidth kernels (warp, correlation lookup) at the bench's shapes: tools/hbm_micro.py"
  timeout 300 python tools/hbm_micro.py gpurun_out/hbm_micro_alg.json > gpurun_out/hbm_micro.log 2>&1; echo "hbm_micro exit $?"
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmcm_$c" -o pmc -- python "$R/tools/hbm_micro.py" /tmp/hbm_micro_alg_$c.json > "$R/gpurun_out/pmcm_$c.log" 2>&1)
    echo "pmc micro $c exit $?"
  done
  python tools/pmc_traffic.py gpurun_out/pmcm_FETCH_SIZE gpurun_out/pmcm_WRITE_SIZE bf16x3 gpurun_out/kernel_traffic.json \
    "tools/hbm_micro.py (3 x the clip's warps at 432x240, RAFT 8 pairs at 864x480 x 20 iterations)" --merge warp,corr_lookup --alg gpurun_out/hbm_micro_alg.json
fi
if has pmc16; then
  echo "== PMC HBM traffic of the MFMA kernels, f16 mode (needs gpurun_out/tuning.json from the bench stage of this visit)"
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && FGT_TUNING_FILE="$R/gpurun_out/tuning.json" timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/gpurun_out/pmcf_$c" -o pmc -- python "$R/bench.py" --precision f16 --steps 1 --warmup 0 --no-cpu-baseline --no-prof --no-fp32-exact --no-c4 > "$R/gpurun_out/pmcf_$c.log" 2>&1)
    echo "pmc f16 $c exit $?"
  done
  python tools/pmc_traffic.py gpurun_out/pmcf_FETCH_SIZE gpurun_out/pmcf_WRITE_SIZE f16 gpurun_out/kernel_traffic.json "bench.py --precision f16 --steps 1 --warmup 0 --no-prof --no-cpu-baseline --no-fp32-exact --no-c4 (prepare pass + 1 step, tiles pre-seeded)"
fi
if has stats; then
  echo "== rocprofv3 kernel stats (same command as the bench headline)"
  rm -rf gpurun_out/prof
  (cd /tmp && FGT_TUNING_FILE="$R/gpurun_out/tuning.json" timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o fgt -- python "$R/bench.py" --steps 2 --warmup 0 --no-cpu-baseline --no-prof --no-fp32-exact --no-f16 --no-c4 > "$R/gpurun_out/rocprof.log" 2>&1)
  echo "rocprof exit: $?"
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats.csv && cut -c1-170 "$f" | head -22
fi
if has stats16; then
  echo "== rocprofv3 kernel stats, f16 mode"
  rm -rf gpurun_out/prof_f16
  (cd /tmp && FGT_TUNING_FILE="$R/gpurun_out/tuning.json" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_f16" -o fgt -- python "$R/bench.py" --precision f16 --steps 2 --warmup 0 --no-cpu-baseline --no-prof --no-c4 --no-fp32-exact > "$R/gpurun_out/rocprof_f16.log" 2>&1)
  echo "rocprof f16 exit: $?"
  f=$(find gpurun_out/prof_f16 -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_f16.csv && cut -c1-170 "$f" | head -12
fi
if has attn; then
  echo "== attention kernels (algorithmic TFLOP/s; fp32 peak 157.3, bf16x3 issues 3x)"
  for p in fp32 bf16x3; do for t in 13 17; do python tools/attn_micro.py --precision $p --t $t 2>&1 | grep attention; done; python tools/attn_micro.py --precision $p --t 17 --spatial 2>&1 | grep attention; python tools/attn_micro.py --precision $p --t 136 --spatial 2>&1 | grep attention; done | tee gpurun_out/attn_micro.log
fi
if has rehearse; then
  echo "== 2-rank rehearsal of bench.py on this single GPU (gloo, collectives staged through the host): strong headline + weak side object"
  FGT_BENCH_SHARE_GPU=1 FGT_BENCH_BACKEND=gloo FGT_TUNING_FILE="$PWD/gpurun_out/tuning.json" timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-fp32-exact > gpurun_out/bench_2rank_rehearsal.log 2>&1
  echo "rehearsal exit: $?"; grep '^{' gpurun_out/bench_2rank_rehearsal.log | cut -c1-1500
fi
if has c5; then
  echo "== BASELINE config #5 clip on one GPU (864x480x160)"
  timeout 900 python bench.py --frames 160 --height 480 --width 864 --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-exact --no-c4 > gpurun_out/bench_c5_1gpu.log 2>&1
  python -c "
import sys,json; d=json.load(open('gpurun_out/bench_detail.json'))
print(d['value'],'fps', d['ms_per_step'],'ms')
for r in d.get('rooflines',[]): print('  ', r['kind'], r['frac'], r.get('algorithmic_tflops', r.get('achieved')),r['unit'], r['kernel_ms_per_step'],'ms/step')
f=d.get('f16')
if f:
    print('f16', f['value'],'fps', f['ms_per_step'],'ms', f.get('composite_vs_headline'))
    for r in f.get('rooflines',[]): print('  ', r['kind'], r['frac'], r.get('algorithmic_tflops', r.get('achieved')),r['unit'], r['kernel_ms_per_step'],'ms/step')
"
fi
if has flow; then
  echo "== flow bench (LAFC / RAFT / config-5 window)"
  timeout 600 python tools/flow_bench.py --cpu > gpurun_out/flow_bench.log 2>&1; tail -5 gpurun_out/flow_bench.log | cut -c1-250
fi
