out=gpurun_out/r03_i
mkdir -p $out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_fft.py -q -m gpu -x -k "two_pass" 2>&1 | tail -15 ) > $out/tests.log
cat $out/tests.log
timeout 900 python tools/bench_sizes.py > $out/bench_sizes.log 2>&1; cp gpurun_out/bench_sizes.json $out/bench_sizes.json
grep -v amdgpu.ids $out/bench_sizes.log | cut -c1-200
