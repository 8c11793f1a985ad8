out=gpurun_out/r03_o
mkdir -p $out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_reg.py tests/test_gpu_batch.py -q -m gpu -x 2>&1 | tail -6 ) > $out/tests.log
cat $out/tests.log
export BENCH_CORR_N=300
( timeout 120 python tools/bench_corr.py; timeout 120 python tools/bench_corr.py --kernel=6; timeout 200 python tools/bench_corr.py --sweep ) 2>&1 | grep -v amdgpu.ids | cut -c1-100 > $out/bench_corr.txt
cat $out/bench_corr.txt
timeout 600 python bench.py --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(json.dumps(j['extra']['reg3d'])); print(json.dumps(j['extra']['reg3d_direct_12dof'])); print(json.dumps(j['extra']['batch_config5']))" > $out/bench_reg.txt
cat $out/bench_reg.txt
