#!/bin/bash
# A/B builds of libapi.so with experimental flags on ONE source:  bash tools/build_variant.sh <name> <file.hip> <flags...>
# -> build/variants/libapi_<name>.so (load it with MIL_LIBAPI_PATH); every other object comes from the regular build.
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
cs=$root/microimagelib_amd/csrc; obj=$root/build/obj; out=$root/build/variants
mkdir -p $out
base="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-fast-math -Wall -Wno-unused-result"
case $src in
  fft2p.hip|fft2p_any_xy.hip|fft2p_any_yz.hip) fl="$base -ffp-contract=fast -fno-slp-vectorize" ;;
  fft.hip) fl="$base -ffp-contract=fast" ;;
  corr_padded.hip|resample_march.hip) fl="$base -fvisibility-inlines-hidden -ffp-contract=off -fno-slp-vectorize -mllvm -disable-vector-combine" ;;
  *) fl="$base -fvisibility-inlines-hidden -ffp-contract=off" ;;
esac
/opt/rocm/bin/hipcc $fl "$@" -c $cs/$src -o $out/$name.$src.o
objs=$(ls $obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc -o $out/libapi_$name.so $objs $out/$name.$src.o -shared -pthread -L/opt/rocm/lib -lrocfft -lamdhip64 -ldl -lz \
  -Wl,-rpath,/opt/rocm/lib -Wl,--no-undefined -Wl,--version-script=$cs/libapi.map -Wl,--exclude-libs,ALL
echo built $out/libapi_$name.so
