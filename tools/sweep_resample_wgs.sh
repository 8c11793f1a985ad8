#!/bin/bash
# tools/sweep_resample_wgs.py on every build/variants/libapi_<pattern>*.so:  bash tools/sweep_resample_wgs.sh <pattern>
for so in build/variants/libapi_${1}*.so; do echo "== $(basename $so)"; MIL_LIBAPI_PATH=$PWD/$so python tools/sweep_resample_wgs.py 2>&1 | grep wgs; done
