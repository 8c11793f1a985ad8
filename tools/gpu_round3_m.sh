out=gpurun_out/r03_m
mkdir -p $out
export TMPDIR=/tmp BENCH_CORR_N=300
for v in "" build/variants/z8/libapi.so build/variants/z32/libapi.so ""; do echo "== lib ${v:-default}"; MIL_LIBAPI_PATH=${v:+$PWD/$v} timeout 120 python tools/bench_corr.py 2>&1 | grep -v amdgpu.ids | cut -c1-90; done > $out/bench_corr_stepsz.txt 2>&1
cat $out/bench_corr_stepsz.txt
