out=gpurun_out/r03_h
mkdir -p $out
export TMPDIR=/tmp
( for n in 0 1 0 1; do MIL_FX_YZ_NARROW=$n timeout 200 python tools/bench_box.py 128 512 512 100; done; MIL_FX_TWOPASS=0 timeout 200 python tools/bench_box.py 128 512 512 100 ) 2>&1 | grep -v amdgpu.ids > $out/bench_box.txt
cat $out/bench_box.txt
