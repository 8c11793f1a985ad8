cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_decon.py -x -q -k ragged 2>&1 | grep -E "passed|failed|Error|assert" | head
timeout 1500 python tools/bench_sizes.py 2>&1 | grep -v amdgpu.ids
