cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_fft.py tests/test_gpu_decon.py tests/test_gpu_batch.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head
for v in 1 2 3; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k.split(' ')[0]: v['ms_per_launch'] for k,v in d['extra']['kernel_families'].items()})"
done
