# scratch: registration cost kernel counters
out=gpurun_out/regprobe
rm -rf $out; mkdir -p $out
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_reg.py -q -m gpu 2>&1 | grep -E "passed|failed|error|FAILED|ERROR" > $out/tests.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/reg_stats -- python tools/bench_corr.py > $out/reg_stats.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/reg_pmc1 -- python tools/bench_corr.py > $out/reg_pmc1.log 2>&1
timeout 600 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum --output-format csv -d $out/reg_pmc2 -- python tools/bench_corr.py > $out/reg_pmc2.log 2>&1
timeout 600 rocprofv3 --pmc TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/reg_pmc3 -- python tools/bench_corr.py > $out/reg_pmc3.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d $out/reg_pmc4 -- python tools/bench_corr.py > $out/reg_pmc4.log 2>&1
find $out -size +20M -delete
cat $out/tests.log; tail -3 $out/reg_pmc3.log $out/reg_pmc4.log
