#!/usr/bin/env python3
"""Condense the rocprofv3 passes of tools/prof_corr.sh: per matrix of tools/bench_corr.py (identity, 3 deg + 2 %, 10 deg,
45 deg; BENCH_CORR_N + 1 launches each) the median kernel duration and one launch's counters.  Prints JSON."""
import collections, csv, glob, json, os, re, sys

NAMES = ["identity", "rot3_scale1.02", "rot10", "rot45"]
N = int(os.environ.get("BENCH_CORR_N", "10")) + 1


def is_cost(name):
    return "corr_" in name and "kernel" in name


def newest(pattern):
    files = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return files[-1] if files else None


def main(src):
    out = {}
    f = newest(os.path.join(src, "trace", "**", "*_kernel_trace.csv"))
    if f:
        rows = [r for r in csv.DictReader(open(f)) if is_cost(r["Kernel_Name"])]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
        out["kernel_us_median"] = {n: sorted(d[i * N:(i + 1) * N])[N // 2] for i, n in enumerate(NAMES) if len(d) >= (i + 1) * N}
        out["kernels"] = {n: re.sub(r"\(anonymous namespace\)::|mil::k::", "", rows[i * N + 1]["Kernel_Name"])[:60]
                          for i, n in enumerate(NAMES) if len(rows) > i * N + 1}
    out["counters"] = {}
    for pm in sorted(glob.glob(os.path.join(src, "pmc*"))):
        if not os.path.isdir(pm):
            continue
        f = newest(os.path.join(pm, "**", "*_counter_collection.csv"))
        if not f:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if is_cost(r["Kernel_Name"]):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out["counters"][k] = {n: v[i * N + 3] for i, n in enumerate(NAMES) if len(v) > i * N + 3}
    c = out["counters"]
    vox = 256 * 512 * 512
    der = {}
    if "FETCH_SIZE" in c:  # KiB; gfx950 tallies 128-B requests at 64 B (MI355X_MICROARCH.md): doubled
        der["hbm_fetch_bytes_2x"] = {n: 2 * 1024 * v for n, v in c["FETCH_SIZE"].items()}
        der["fetch_over_algorithmic_2R"] = {n: round(2 * 1024 * v / (2 * 4 * vox), 3) for n, v in c["FETCH_SIZE"].items()}
    if "SQ_INSTS_VALU" in c:
        der["valu_per_voxel_wave"] = {n: round(v / (vox / 64), 2) for n, v in c["SQ_INSTS_VALU"].items()}
        der["vmem_per_voxel_wave"] = {n: round(v / (vox / 64), 2) for n, v in c["SQ_INSTS_VMEM_RD"].items()}
    out["derived"] = der
    json.dump(out, open(os.path.join(src, "summary.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
