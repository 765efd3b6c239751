timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_split_gpu.py tests/test_flow_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -3
for l in qkv8_k32 qkv8; do for t in 128x128x8 256x256x8il; do python tools/conv_micro.py --layer $l --tile $t --split planes --reps 10 2>&1 | grep -v amdgpu; done; done
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'],'fps', d['ms_per_step'],'ms', d['roofline']['algorithmic_tflops'], d['roofline']['frac'], d['output_checksum'])"
