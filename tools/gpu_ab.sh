cd /root/repo
for v in 256 512 256 512; do
MIL_FX_COLNT=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], {k.split(' ')[0]: v['ms_per_launch'] for k,v in d['extra']['kernel_families'].items()})"
done
MIL_FX_COLNT=512 timeout 600 python -m pytest tests/test_gpu_fft.py -x -q 2>&1 | grep -E "passed|failed"
python -c "import __graft_entry__ as g; g.smoke()"
