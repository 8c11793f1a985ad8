# counters of the yz kernel at the benchmark box (bench.py --steps 5): bash tools/prof_yz_lds.sh tag
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/yzlds_$1; mkdir -p $out; rm -rf $out/p1 $out/p2 $out/tr
C="python bench.py --steps 5 --warmup 1 --clock-warm-steps 0 --no-cpu-baseline --no-extra"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr -- $C > $out/tr.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $out/p1 -- $C > $out/p1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/p2 -- $C > $out/p2.log 2>&1
python - "$out" <<'PY'
import csv, glob, collections, os, sys
out = sys.argv[1]
f = sorted(glob.glob(f"{out}/tr/*/*kernel_stats.csv"), key=os.path.getsize)[-1]
for r in csv.DictReader(open(f)):
    if "yz_" in r["Name"]: print(r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3)
for d in ("p1", "p2"):
    f = sorted(glob.glob(f"{out}/{d}/*/*counter_collection.csv"), key=os.path.getsize)[-1]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "yz_" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d, {c: round(sum(v) / len(v)) for c, v in sorted(acc.items())})
PY
