# round 3, first GPU call: tests, bench, TA probe, RCCL with two ranks on the one device
out=gpurun_out/r03_a
mkdir -p $out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 ) > $out/tests.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
timeout 120 build/probe_ta > $out/probe_ta.txt 2>&1
( HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --share-gpu --dist-backend nccl --steps 20 --no-cpu-baseline ) > $out/bench_2ranks_nccl_shared_gpu.json 2> $out/bench_2ranks_nccl.err
tail -5 $out/bench_2ranks_nccl.err
cat $out/tests.log $out/probe_ta.txt
cat $out/bench.json
