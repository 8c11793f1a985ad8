cd /root/repo
for i in 1 2; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('events   ', d['value'], d['ms_per_step'])"
MIL_BENCH_NO_EVENTS=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no events', d['value'], d['ms_per_step'])"
done
python bench.py --steps 100 --warmup 3 --no-cpu-baseline --no-extra | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('events K=100', d['value'], d['ms_per_step'])"
