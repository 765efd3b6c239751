export TMPDIR=/tmp; mkdir -p gpurun_out
summ() { python - "$1" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
ln=[r for r in d["rooflines"] if r["kind"]=="layernorm"][0]
print(d["value"],"fps", "checksum", d.get("output_checksum"), "LN", ln["frac"], ln["kernel_ms_per_step"])
PY
}
for v in 2 1 3 2 1 3; do
  if [ $v = 2 ]; then unset FGT_HIP_LIB; else export FGT_HIP_LIB=$PWD/fgt_amd/lib/libfgt_hip_ln$v.so; fi
  echo "== LN rows per wavefront $v"; timeout 600 python bench.py --steps 3 --warmup 1 --no-c4 --no-f16 --no-fp32-exact --no-cpu-baseline > gpurun_out/bench_ln_$v.log 2>&1; summ gpurun_out/bench_ln_$v.log
done
