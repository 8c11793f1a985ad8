cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_fft.py tests/test_gpu_decon.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
