cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/corrprof
rocprofv3 -L > gpurun_out/corrprof/counters.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/corrprof/stats -- python tools/bench_corr.py > gpurun_out/corrprof/stats.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/corrprof/pmc1 -- python tools/bench_corr.py > gpurun_out/corrprof/pmc1.log 2>&1
timeout 600 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum --output-format csv -d gpurun_out/corrprof/pmc2 -- python tools/bench_corr.py > gpurun_out/corrprof/pmc2.log 2>&1
ls -la gpurun_out/corrprof/*/*/ | head -30
