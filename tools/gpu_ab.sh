cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_fft.py tests/test_gpu_decon.py tests/test_gpu_batch.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head
for g in 1 0; do
echo "MIL_FX_GRAPH=$g"
MIL_FX_GRAPH=$g python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
MIL_FX_GRAPH=$g MIL_BENCH_NO_EVENTS=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no events', d['value'], d['ms_per_step'])"
MIL_FX_GRAPH=$g timeout 900 python tools/bench_sizes.py 2>&1 | grep -E "\(64, 64, 64\)|\(128, 128, 128\)|\(256, 256, 256\)|\(192, 192" | cut -c1-110
done
