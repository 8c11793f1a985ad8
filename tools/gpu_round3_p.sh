out=gpurun_out/r03_p
mkdir -p $out
export TMPDIR=/tmp BENCH_CORR_N=200
( for k in 3 6 5 3 6; do timeout 200 python tools/bench_corr.py --sweep --kernel=$k; done ) 2>&1 | grep -v amdgpu.ids | cut -c1-60 > $out/bench_corr_sweep.txt
cat $out/bench_corr_sweep.txt
