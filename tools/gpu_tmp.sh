timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_split_gpu.py tests/test_fgt_gpu.py -q -x -p no:cacheprovider -k "attention or attn or fgt" 2>&1 | tail -3
python tools/attn_micro.py --precision bf16x3 --t 17 2>&1 | grep attention; python tools/attn_micro.py --precision bf16x3 --t 17 --spatial 2>&1 | grep attention
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'],'fps', d['ms_per_step'],'ms', d['output_checksum'])"
