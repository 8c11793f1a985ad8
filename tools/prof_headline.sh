cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/hl; mkdir -p $out; rm -rf $out/stats $out/pmc_fetch $out/pmc_write
B20="python bench.py --steps 20 --warmup 3 --clock-warm-steps 0 --no-cpu-baseline --no-extra"
B5="python bench.py --steps 5 --warmup 1 --clock-warm-steps 0 --no-cpu-baseline --no-extra"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $B20 > $out/prof_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B5 > $out/prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $B5 > $out/prof_write.log 2>&1
python tools/rocprof_summary.py $out $out/summary > $out/summary.log 2>&1
cat $out/pmc_traffic.json | head -8; grep -i "yz_conv\|xy_kernel" $out/summary/kernel_stats.csv | cut -c1-200
