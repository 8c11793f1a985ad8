# rocprofv3 passes over tools/bench_corr.py for one forced cost kernel:  bash tools/prof_corr.sh <tag> <kernel>
tag=${1:-corr}; kern=${2:--1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp BENCH_CORR_N=10
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python tools/bench_corr.py --kernel=$kern > $out/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python tools/bench_corr.py --kernel=$kern > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/pmc1 -- python tools/bench_corr.py --kernel=$kern > $out/pmc1.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc2 -- python tools/bench_corr.py --kernel=$kern > $out/pmc2.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD --output-format csv -d $out/pmc4 -- python tools/bench_corr.py --kernel=$kern > $out/pmc4.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $out/pmc3 -- python tools/bench_corr.py --kernel=$kern > $out/pmc3.log 2>&1
find $out -size +20M -delete
python tools/prof_corr_summary.py $out
