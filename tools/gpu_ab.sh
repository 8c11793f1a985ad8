cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_reg.py -x -q -k full_size 2>&1 | grep -E "passed|failed|^E " | head -30
