out=gpurun_out/r03_b
mkdir -p $out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_reg.py -q -m gpu -x 2>&1 | tail -15 ) > $out/tests.log
export BENCH_CORR_N=300
for v in 0 1 2 3 1 0; do echo "== variant $v"; timeout 120 python tools/bench_corr.py --variant=$v; done > $out/bench_corr_variants.txt 2>&1
( timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --force-dist --dist-backend nccl --steps 20 --no-cpu-baseline ) > $out/bench_1rank_nccl_forced.json 2> $out/bench_1rank_nccl.err
cat $out/tests.log $out/bench_corr_variants.txt; tail -3 $out/bench_1rank_nccl.err; cut -c1-400 $out/bench_1rank_nccl_forced.json
