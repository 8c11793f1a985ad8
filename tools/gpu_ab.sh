cd /root/repo
timeout 900 python -m pytest tests/test_gpu_cli.py -x -q 2>&1 | tail -15
