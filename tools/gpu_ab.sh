cd /root/repo
timeout 900 python -m pytest tests/test_gpu_reg.py tests/test_gpu_kernels.py tests/test_gpu_cli.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head
echo "--- march"; timeout 300 python tools/bench_corr.py 2>/dev/null
echo "--- direct"; MIL_CORR_MODE=direct timeout 300 python tools/bench_corr.py 2>/dev/null
