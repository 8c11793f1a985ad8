# rocprofv3 passes over the two-pass RL step:  bash tools/prof_twopass.sh <tag> [MIL_FX_TWOPASS value]
tag=${1:-twopass}; mode=${2:-3}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp MIL_FX_TWOPASS=$mode
B="python bench.py --steps 5 --warmup 1 --clock-warm-steps 0 --no-cpu-baseline --no-extra"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B > $out/trace.log 2>&1
if [ -z "$SKIP_TRAFFIC" ]; then
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B > $out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $B > $out/pmc_write.log 2>&1
fi
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC --output-format csv -d $out/pmc1 -- $B > $out/pmc1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU --output-format csv -d $out/pmc2 -- $B > $out/pmc2.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $out/pmc3 -- $B > $out/pmc3.log 2>&1
find $out -size +20M -delete
python tools/prof_twopass_summary.py $out
