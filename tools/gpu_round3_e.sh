out=gpurun_out/r03_e
mkdir -p $out
export TMPDIR=/tmp
export BENCH_CORR_N=300
for args in "--kernel=5" "--kernel=4" "--kernel=1" "--kernel=2" "--kernel=5"; do echo "== $args"; timeout 120 python tools/bench_corr.py $args 2>&1 | grep -v amdgpu.ids; done > $out/bench_corr_variants.txt 2>&1
echo "== --kernel=5 MIL_CORR_EARLY=1" >> $out/bench_corr_variants.txt
MIL_CORR_EARLY=1 timeout 120 python tools/bench_corr.py --kernel=5 2>&1 | grep -v amdgpu.ids >> $out/bench_corr_variants.txt
cat $out/bench_corr_variants.txt
