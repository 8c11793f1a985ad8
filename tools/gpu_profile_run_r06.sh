# Round-6 evidence: bench (headline + extras + CPU baseline; K = 100 and the driver's K = 20), the self-launched 2-rank run and
# the forced one-rank RCCL run, rocprofv3 stats and PMC passes of the RL step (-> profiles/pmc_traffic.json), of the AUXILIARY
# kernels (FETCH_SIZE / WRITE_SIZE / VALU activity: tools/prof_twopass_summary.py -> aux_pmc.json) and of the registration
# cost kernels, the host gap of a registration, the box sweep.
#   usage (on the GPU box): bash tools/gpu_profile_run_r06.sh [tag]
tag=${1:-r06}
out=gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B20="python bench.py --steps 20 --warmup 3 --clock-warm-steps 0 --no-cpu-baseline --no-extra"
B5="python bench.py --steps 5 --warmup 1 --clock-warm-steps 0 --no-cpu-baseline --no-extra"
rm -rf $out/stats $out/pmc_fetch $out/pmc_write
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $B20 > $out/prof_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B5 > $out/prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $B5 > $out/prof_write.log 2>&1
find $out -size +20M -delete
python tools/rocprof_summary.py $out $out/summary > $out/summary.log 2>&1   # writes $out/pmc_traffic.json (-> profiles/pmc_traffic.json)
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
timeout 600 python bench.py --steps 20 --no-cpu-baseline > $out/bench_steps20.json 2> $out/bench_steps20.err
timeout 300 python bench.py --no-cpu-baseline --no-extra --backend rocfft > $out/bench_rocfft_backend.json 2> $out/bench_rocfft.err
timeout 600 python bench.py --gpus 2 --share-gpu --dist-backend gloo --steps 20 --no-cpu-baseline --batch-time-points-per-gpu 4 > $out/bench_2ranks_self_launched_gloo.json 2> $out/bench_2ranks.err
( timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --force-dist --dist-backend nccl --steps 20 --no-cpu-baseline --batch-time-points-per-gpu 4 ) > $out/bench_1rank_rccl_forced.json 2> $out/bench_1rank_rccl.err
# occupancy / wait split of the two-pass kernels
SKIP_TRAFFIC=1 bash tools/prof_twopass.sh $tag/twopass 3 > $out/twopass.log 2>&1
# auxiliary kernels: durations + FETCH_SIZE / WRITE_SIZE / VALU activity per kernel
A="python tools/bench_aux.py"
rm -rf $out/aux
mkdir -p $out/aux
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/aux/trace -- $A > $out/aux/trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/aux/pmc_fetch -- $A > $out/aux/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/aux/pmc_write -- $A > $out/aux/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/aux/pmc1 -- $A > $out/aux/pmc1.log 2>&1
find $out -size +20M -delete
python tools/prof_twopass_summary.py $out/aux > $out/aux/summary.log 2>&1
# registration cost kernels (tools/reg_profile_summary.py condenses these) and the host gap of a whole registration
rm -rf $out/reg_stats $out/reg_pmc1 $out/reg_pmc2 $out/reg_pmc3 $out/reg_gap
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/reg_stats -- python tools/bench_corr.py > $out/reg_stats.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/reg_pmc1 -- python tools/bench_corr.py > $out/reg_pmc1.log 2>&1
timeout 600 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum --output-format csv -d $out/reg_pmc2 -- python tools/bench_corr.py > $out/reg_pmc2.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/reg_pmc3 -- python tools/bench_corr.py > $out/reg_pmc3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/reg_gap -- python tools/reg_gap.py > $out/reg_gap.log 2>&1
find $out -size +20M -delete
python tools/reg_profile_summary.py $out $out/summary > $out/reg_summary.log 2>&1
python tools/reg_gap_summary.py $out/reg_gap $out/reg_gap.log > $out/reg_host_gap.txt 2>&1
timeout 1500 python tools/bench_sizes.py > $out/bench_sizes.log 2>&1; cp gpurun_out/bench_sizes.json $out/bench_sizes.json
( head -4 profiles/r05/resample_ab.txt; python tools/bench_resample_ab.py ) > $out/resample_ab.txt 2>&1
python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|error" | tail -3 > $out/gpu_tests.log
cut -c1-300 $out/bench.json; cut -c1-200 $out/bench_steps20.json; cat $out/reg_host_gap.txt | tail -5
