#!/bin/bash
# rocprofv3 kernel stats of the RL step for one library build:  bash tools/prof_variant.sh <libapi.so or ""> <tag>
so=$1; tag=$2; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
[ -n "$so" ] && export MIL_LIBAPI_PATH=$PWD/$so
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench.py --steps 20 --warmup 3 --clock-warm-steps 30 --no-cpu-baseline --no-extra > $out/trace.log 2>&1
f=$(find $out/trace -name "*kernel_stats.csv" | head -n1)
echo "== $tag"; python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"])>1: print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us  {r["Percentage"]}%')
PY
find $out -size +5M -delete
