out=gpurun_out/r03_d
mkdir -p $out
export TMPDIR=/tmp
export BENCH_CORR_N=300
for args in "--variant=1" "--kernel=3 --variant=5" "--kernel=3 --variant=6" "--kernel=3 --variant=7" "--kernel=3 --variant=8" "--kernel=3 --variant=9" "--variant=1"; do echo "== $args"; timeout 120 python tools/bench_corr.py $args 2>&1 | grep -v amdgpu.ids; done > $out/bench_corr_variants.txt 2>&1
cat $out/bench_corr_variants.txt
