# rocprofv3 kernel stats of the two-pass kernels against the classic sequence on given boxes (tools/twopass_any_check.py)
#   usage (GPU box): bash tools/prof_any.sh "256,384,384;128,320,448" tag
boxes=${1:-"256,384,384;128,320,448"}
tag=${2:-any}
out=gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf $out/trace
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python tools/twopass_any_check.py --boxes="$boxes" --iters=20 > $out/trace.log 2>&1
f=$(ls -S $out/trace/*/*_kernel_stats.csv | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if "mil::fx" in n:
        print(f'{n.replace("mil::fx::(anonymous namespace)::", "").replace("void ", "")[:90]:90s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:8.1f} us')
PY
tail -3 $out/trace.log
