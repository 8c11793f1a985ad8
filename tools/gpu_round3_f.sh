out=gpurun_out/r03_f
mkdir -p $out
export TMPDIR=/tmp
( timeout 900 python tools/bench_cli_batch.py --time-points 8 --reg-mode 3 --workers 0,0; timeout 900 python tools/bench_cli_batch.py --time-points 8 --reg-mode 2 --workers 0,0; timeout 900 python tools/bench_cli_batch.py --time-points 9 --reg-mode 3 --workers 0,0,0 ) > $out/bench_cli_batch_workers.json 2> $out/bench_cli_batch_workers.err
cat $out/bench_cli_batch_workers.json; tail -3 $out/bench_cli_batch_workers.err
