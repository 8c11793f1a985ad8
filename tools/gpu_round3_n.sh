out=gpurun_out/r03_n
mkdir -p $out
export TMPDIR=/tmp BENCH_CORR_N=300
for v in "" build/variants/my/libapi.so "" build/variants/my/libapi.so; do echo "== lib ${v:-default} --kernel=3"; MIL_LIBAPI_PATH=${v:+$PWD/$v} timeout 120 python tools/bench_corr.py --kernel=3 2>&1 | grep -v amdgpu.ids | cut -c1-150; done > $out/bench_corr_marchy.txt 2>&1
cat $out/bench_corr_marchy.txt
