out=gpurun_out/r03_j
mkdir -p $out
export TMPDIR=/tmp
( for tp in 0 2 3; do MIL_FX_TWOPASS=$tp timeout 300 python tools/bench_box.py 256 1024 1024 30; MIL_FX_TWOPASS=$tp timeout 300 python tools/bench_box.py 256 1024 512 50; done ) 2>&1 | grep -v amdgpu.ids > $out/bench_box.txt
cat $out/bench_box.txt
