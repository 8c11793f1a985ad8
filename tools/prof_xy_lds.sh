# LDS / wait counters of the xy kernels per NH (boxes 128 x 512 x nx): bash tools/prof_xy_lds.sh "512;576;768;1024"
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/xylds; mkdir -p $out; rm -rf $out/p1 $out/p2 $out/tr
B=""; for nx in ${1//;/ }; do B="$B;128,512,$nx"; done; B=${B#;}
C="python tools/xy_any_probe.py --boxes=$B --iters=10"
export MIL_FX_ANY=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr -- $C > $out/tr.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $out/p1 -- $C > $out/p1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/p2 -- $C > $out/p2.log 2>&1
python - <<'PY'
import csv, glob, collections
for d in ("p1", "p2"):
    f = sorted(glob.glob(f"gpurun_out/xylds/{d}/*/*counter_collection.csv"), key=lambda p: -__import__("os").path.getsize(p))[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "xy_kernel<2" in n:
            acc[n.split("xy_kernel")[1][:20]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print(d, k, {c: round(sum(v) / len(v)) for c, v in sorted(acc[k].items())})
PY
