# Round-4 evidence: GPU tests, bench (headline + extras + CPU baseline), the self-launched 2-rank run, rocprofv3 stats and PMC
# passes (RL step, auxiliary kernels, registration cost kernel), box sweep, batch and command-line figures.
#   usage (on the GPU box): bash tools/gpu_profile_run_r04.sh [tag]
tag=${1:-r04_final}
out=gpurun_out/$tag
mkdir -p $out
cd /root/repo
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error|FAILED|ERROR" ) > $out/tests.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
timeout 300 python bench.py --no-cpu-baseline --no-extra --backend rocfft > $out/bench_rocfft_backend.json 2> $out/bench_rocfft.err
# `bench.py --gpus N` from a plain shell: two ranks on this box's one GPU, collectives over gloo (RCCL refuses two ranks on one device)
timeout 600 python bench.py --gpus 2 --share-gpu --dist-backend gloo --steps 20 --no-cpu-baseline > $out/bench_2ranks_self_launched_gloo.json 2> $out/bench_2ranks.err
( timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --force-dist --dist-backend nccl --steps 20 --no-cpu-baseline --batch-tiff-dir "" ) > $out/bench_1rank_rccl_forced.json 2> $out/bench_1rank_rccl.err
B="python bench.py --steps 20 --warmup 3 --clock-warm-steps 0 --no-cpu-baseline --no-extra"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $B > $out/prof_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python bench.py --steps 5 --warmup 1 --clock-warm-steps 0 --no-cpu-baseline --no-extra > $out/prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python bench.py --steps 5 --warmup 1 --clock-warm-steps 0 --no-cpu-baseline --no-extra > $out/prof_write.log 2>&1
# occupancy / wait split of the two-pass kernels (tools/prof_twopass_summary.py -> twopass/summary.json)
SKIP_TRAFFIC=1 bash tools/prof_twopass.sh $tag/twopass 3 > $out/twopass.log 2>&1
# auxiliary kernels (resample, projections, pad / crop, 3-D MIP series): kernel names and durations as rocprofv3 sees them
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/aux_stats -- python tools/bench_aux.py > $out/aux_stats.log 2>&1
find $out -size +20M -delete
cat $out/tests.log; cut -c1-400 $out/bench.json
timeout 900 python tools/bench_sizes.py > $out/bench_sizes.log 2>&1; cp gpurun_out/bench_sizes.json $out/bench_sizes.json
# BASELINE.json configs[4] on one GPU
( timeout 300 python tools/bench_batch.py --time-points 8; timeout 300 python tools/bench_batch.py --time-points 8 --lanes 2; timeout 300 python tools/bench_batch.py --time-points 8 --reg-mode 2; timeout 300 python tools/bench_batch.py --time-points 8 --reg-mode 1; timeout 400 python tools/bench_batch.py --time-points 8 --tiff /tmp/mil_batch_tiff --overlap ) > $out/bench_batch.json 2> $out/bench_batch.err
( timeout 900 python tools/bench_cli_batch.py --time-points 16 --reg-mode 3 --workers 0,0 --distinct 4; timeout 900 python tools/bench_cli_batch.py --time-points 16 --reg-mode 2 --workers 0,0 --distinct 4 ) > $out/bench_cli_batch.json 2> $out/bench_cli_batch.err
# registration cost kernel (tools/reg_profile_summary.py condenses these)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/reg_stats -- python tools/bench_corr.py > $out/reg_stats.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/reg_pmc1 -- python tools/bench_corr.py > $out/reg_pmc1.log 2>&1
timeout 600 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum --output-format csv -d $out/reg_pmc2 -- python tools/bench_corr.py > $out/reg_pmc2.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/reg_pmc3 -- python tools/bench_corr.py > $out/reg_pmc3.log 2>&1
find $out -size +20M -delete
