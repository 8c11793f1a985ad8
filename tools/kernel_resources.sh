#!/bin/bash
# VGPRs / scratch / occupancy / LDS of every kernel of one .hip file:  bash tools/kernel_resources.sh fft2p.hip [extra flags]
# (run from microimagelib_amd/csrc; flags as in the Makefile's FFTFLAGS)
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=fast -fno-fast-math -fno-slp-vectorize "$@" \
    -Rpass-analysis=kernel-resource-usage -c $f -o /dev/null 2>&1 |
  grep "Function Name\|VGPRs:\|ScratchSize\|Occupancy\|LDS Size" | sed 's/.*remark: *//; s/\[-Rpass.*//' | paste - - - - - |
  sed 's/Function Name: _ZN3mil2fx12_GLOBAL__N_1//; s/ScratchSize \[bytes\/lane\]/Scratch/; s/Occupancy \[waves\/SIMD\]/Occ/; s/LDS Size \[bytes\/block\]/LDS/'
