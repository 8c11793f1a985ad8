# scratch: slab kernel counters
out=gpurun_out/regprobe
rm -rf $out; mkdir -p $out
cd /root/repo
export TMPDIR=/tmp
export BENCH_CORR_N=4
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/reg_stats -- python tools/bench_corr.py > $out/reg_stats.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/reg_pmc1 -- python tools/bench_corr.py > $out/reg_pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD TA_BUSY_avr SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d $out/reg_pmc2 -- python tools/bench_corr.py > $out/reg_pmc2.log 2>&1
find $out -size +20M -delete
tail -n 3 $out/reg_pmc2.log
