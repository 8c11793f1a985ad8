out=gpurun_out/r03_l
mkdir -p $out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_reg.py -q -m gpu -x 2>&1 | tail -4 ) > $out/tests.log
cat $out/tests.log
export BENCH_CORR_N=300
for w in 1 0 1 0; do echo "== MIL_CORR_WAVES5=$w"; MIL_CORR_WAVES5=$w timeout 120 python tools/bench_corr.py 2>&1 | grep -v amdgpu.ids | cut -c1-90; done > $out/bench_corr_waves.txt 2>&1
cat $out/bench_corr_waves.txt
