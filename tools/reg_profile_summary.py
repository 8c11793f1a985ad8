#!/usr/bin/env python3
"""Condense the rocprofv3 runs of tools/bench_corr.py (gpurun_out/<tag>/reg_*) into profiles/<tag>/reg_cost_kernel.json.

    python tools/reg_profile_summary.py gpurun_out/r01_final profiles/r01_final
"""
import collections, csv, glob, json, os, re, sys

NAMES = ["identity", "rot3_scale1.02", "rot10", "rot45"]  # order and 51 launches each: tools/bench_corr.py


def is_cost(name):
    return "corr_" in name and "kernel" in name  # corr_kernel, corr_padded_kernel, corr_rowpair_kernel, corr_pair_kernel


def newest(pattern):
    files = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return files[-1] if files else None


def main(src, dst):
    out = {"command": "rocprofv3 ... -- python tools/bench_corr.py (4 matrices x 51 evaluations of mil_corr_sums_padded at "
                      "512x512x256); counters = one launch of each matrix (the 11th)"}
    f = newest(os.path.join(src, "reg_stats", "**", "*_kernel_trace.csv"))
    rows = [r for r in csv.DictReader(open(f)) if is_cost(r["Kernel_Name"])]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    out["kernel_us_median"] = {n: sorted(d[i * 51:(i + 1) * 51])[25] for i, n in enumerate(NAMES)}
    out["kernels"] = {n: re.search(r"corr_\w+(<\w+>)?", rows[i * 51 + 10]["Kernel_Name"]).group(0) for i, n in enumerate(NAMES)}
    out["counters"] = {}
    for pm in ("reg_pmc1", "reg_pmc2", "reg_pmc3"):
        f = newest(os.path.join(src, pm, "**", "*_counter_collection.csv"))
        if not f:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if is_cost(r["Kernel_Name"]):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out["counters"][k] = {n: v[i * 51 + 10] for i, n in enumerate(NAMES) if len(v) > i * 51 + 10}
    vox = 256 * 512 * 512
    c = out["counters"]
    if "SQ_INSTS_VALU" in c:
        out["derived_identity"] = {
            "valu_insts_per_voxel_wave": c["SQ_INSTS_VALU"]["identity"] / (vox / 64),
            "vmem_insts_per_voxel_wave": c["SQ_INSTS_VMEM_RD"]["identity"] / (vox / 64),
            "l1_accesses_per_vmem_inst": c["TCP_TOTAL_CACHE_ACCESSES_sum"]["identity"] / c["SQ_INSTS_VMEM_RD"]["identity"],
            "algorithmic_bytes": 2 * 4 * vox}
    os.makedirs(dst, exist_ok=True)
    json.dump(out, open(os.path.join(dst, "reg_cost_kernel.json"), "w"), indent=1)
    print(json.dumps(out["kernel_us_median"]), json.dumps(out.get("derived_identity")))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
