cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_batch.py -x -q 2>&1 | grep -E "passed|failed|^E " | head -30
