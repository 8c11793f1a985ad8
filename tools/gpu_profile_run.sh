set -x
mkdir -p gpurun_out/r01_fused
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_decon.py tests/test_gpu_fft.py -x -q 2>&1 | tail -8 > gpurun_out/r01_fused/tests.log
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/r01_fused/bench_fused.json 2> gpurun_out/r01_fused/bench_fused.err
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --backend rocfft > gpurun_out/r01_fused/bench_rocfft.json 2> gpurun_out/r01_fused/bench_rocfft.err
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01_fused/stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/r01_fused/prof_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r01_fused/pmc_fetch -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/r01_fused/prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r01_fused/pmc_write -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/r01_fused/prof_write.log 2>&1
find gpurun_out/r01_fused -name "*.csv" | xargs ls -la
# keep the merged output small: drop per-dispatch traces > 20 MB
find gpurun_out/r01_fused -size +20M -delete
