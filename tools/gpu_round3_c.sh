out=gpurun_out/r03_c
mkdir -p $out
export TMPDIR=/tmp
timeout 200 build/probe_ta > $out/probe_ta.txt 2>&1
( timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "padded" 2>&1 | tail -5 ) > $out/tests.log
export BENCH_CORR_N=300
for args in "--kernel=3 --variant=1" "--kernel=3 --variant=4" "--kernel=5 --variant=1" "--kernel=3 --variant=4" "--kernel=3 --variant=1"; do echo "== $args"; timeout 120 python tools/bench_corr.py $args; done > $out/bench_corr_variants.txt 2>&1
cat $out/probe_ta.txt $out/tests.log $out/bench_corr_variants.txt
