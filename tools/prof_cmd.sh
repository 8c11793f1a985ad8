# rocprofv3 kernel stats (mil::fx kernels only) of an arbitrary python command line:  bash tools/prof_cmd.sh tag python tools/x.py ...
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf $out/trace
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- "$@" > $out/trace.log 2>&1
f=$(ls -S $out/trace/*/*_kernel_stats.csv | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "mil::" in n:
        print(f'{n.replace("mil::fx::(anonymous namespace)::", "").replace("void ", "")[:90]:90s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:8.1f} us')
PY
grep "it/s" $out/trace.log
