#!/bin/bash
# bench_box on every build/variants/libapi_<pattern>*.so, interleaved twice:  bash tools/ab_variants.sh <pattern> [NZ NY NX]
pat=$1; shift; box=${@:-256 512 512}
mkdir -p gpurun_out
for rep in 1 2; do
  for so in build/variants/libapi_${pat}*.so; do
    echo -n "$(basename $so): "
    MIL_LIBAPI_PATH=$PWD/$so timeout 300 python tools/bench_box.py $box 100 2>&1 | tail -n1
  done
done
