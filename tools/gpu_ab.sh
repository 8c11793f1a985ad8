cd /root/repo
timeout 900 python -m pytest tests/test_gpu_reg.py tests/test_gpu_kernels.py tests/test_gpu_batch.py -x -q 2>&1 | grep -E "passed|failed|^E " | head -20
