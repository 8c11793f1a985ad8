cd /root/repo
timeout 900 python -m pytest tests/test_gpu_reg.py tests/test_gpu_kernels.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head
timeout 300 python tools/bench_corr.py 2>/dev/null
