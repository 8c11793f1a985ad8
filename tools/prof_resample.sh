cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_resample -- python tools/bench_resample_ab.py > gpurun_out/prof_resample.log 2>&1
f=$(ls gpurun_out/prof_resample/*/*kernel_stats.csv | head -1)
head -12 $f
