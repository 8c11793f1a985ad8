out=gpurun_out/r03_k
mkdir -p $out
export TMPDIR=/tmp
for lanes in 1 2 3 1 2; do
  timeout 600 python bench.py --steps 20 --no-cpu-baseline --batch-lanes $lanes --batch-time-points-per-gpu 6 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('lanes', j['extra']['batch_config5']['time_points_in_flight_per_gpu'], j['extra']['batch_config5']['time_points_per_s'], 'tp/s', j['value'], 'it/s')"
done > $out/batch_lanes.txt 2>&1
cat $out/batch_lanes.txt
