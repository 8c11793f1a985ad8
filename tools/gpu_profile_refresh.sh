# Re-run the parts of the round's evidence that depend on the kernel sources (after an edit): GPU tests, bench line, rocprofv3
# stats + FETCH_SIZE / WRITE_SIZE passes (profiles/pmc_traffic.json is keyed on the sources' hashes), auxiliary-kernel stats.
#   usage (on the GPU box): bash tools/gpu_profile_refresh.sh [tag]
tag=${1:-r04_final}
out=gpurun_out/$tag
mkdir -p $out
cd /root/repo
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error|FAILED|ERROR" ) > $out/tests.log
B="python bench.py --steps 20 --warmup 3 --clock-warm-steps 0 --no-cpu-baseline --no-extra"
rm -rf $out/stats $out/pmc_fetch $out/pmc_write $out/aux_stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $B > $out/prof_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python bench.py --steps 5 --warmup 1 --clock-warm-steps 0 --no-cpu-baseline --no-extra > $out/prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python bench.py --steps 5 --warmup 1 --clock-warm-steps 0 --no-cpu-baseline --no-extra > $out/prof_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/aux_stats -- python tools/bench_aux.py > $out/aux_stats.log 2>&1
find $out -size +20M -delete
python tools/rocprof_summary.py $out /tmp/refresh_profiles > /dev/null 2>&1   # refreshes profiles/pmc_traffic.json in this copy
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
timeout 600 python bench.py --gpus 2 --share-gpu --dist-backend gloo --steps 20 --no-cpu-baseline > $out/bench_2ranks_self_launched_gloo.json 2> $out/bench_2ranks.err
( timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --force-dist --dist-backend nccl --steps 20 --no-cpu-baseline ) > $out/bench_1rank_rccl_forced.json 2> $out/bench_1rank_rccl.err
# registration cost kernel (tools/reg_profile_summary.py condenses these)
rm -rf $out/reg_stats $out/reg_pmc1 $out/reg_pmc2 $out/reg_pmc3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/reg_stats -- python tools/bench_corr.py > $out/reg_stats.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/reg_pmc1 -- python tools/bench_corr.py > $out/reg_pmc1.log 2>&1
timeout 600 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum --output-format csv -d $out/reg_pmc2 -- python tools/bench_corr.py > $out/reg_pmc2.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/reg_pmc3 -- python tools/bench_corr.py > $out/reg_pmc3.log 2>&1
( timeout 300 python tools/bench_batch.py --time-points 8; timeout 300 python tools/bench_batch.py --time-points 8 --lanes 2; timeout 300 python tools/bench_batch.py --time-points 8 --reg-mode 2; timeout 300 python tools/bench_batch.py --time-points 8 --reg-mode 1 ) > $out/bench_batch.json 2> $out/bench_batch.err
find $out -size +20M -delete
cat $out/tests.log; cut -c1-300 $out/bench.json; cp profiles/pmc_traffic.json $out/pmc_traffic.json
