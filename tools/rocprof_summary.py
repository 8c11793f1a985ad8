#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/<tag>/...) into the small files kept under profiles/<tag>/.

    python tools/rocprof_summary.py gpurun_out/r01_fused profiles/r01_fused

  stats/      rocprofv3 --kernel-trace --stats   -> kernel_stats.csv (names shortened)
  pmc_fetch/  rocprofv3 --pmc FETCH_SIZE         -> pmc_per_kernel.json: mean per launch, KiB
  pmc_write/  rocprofv3 --pmc WRITE_SIZE            FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM: gfx950 tallies the
                                                   128-B requests of a 16 B/lane stream at 64 B), WRITE_SIZE is exact
and refreshes profiles/pmc_traffic.json (bytes per launch of our kernels, read by bench.py for roofline.traffic).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

OURS = {  # kernel-name pattern -> bench.py family key (config size 512x512x256)
    "fused": {r"x_kernel<256, 3,": "x_update", r"x_kernel<256, 2,": "x_ratio", r"col_kernel<512, 512, [01]>": "y_pass",
              r"col_kernel<256, 256, 2>": "z_conv",
              # two passes per convolution (csrc/fft2p.hip)
              r"yz_conv_kernel|yz_lds_kernel<512, 256,": "yz_conv", r"xy_kernel<2(, 512, 256)?(, (true|false))?>": "xy_ratio",
              r"xy_kernel<3(, 512, 256)?(, (true|false))?>": "xy_update"},
    "rocfft": {r"ew_inplace2_kernel<.*UpdateOp": "rl_update", r"ew_inplace2_kernel<.*RatioOp": "rl_ratio",
               r"cmul_kernel": "cmul_otf"},
}


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return name if len(name) < 160 else name[:157] + "..."


def counter_means(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    files = sorted(glob.glob(os.path.join(path, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:  # newest run only (gpurun merges into an existing directory)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                a = acc[short(r["Kernel_Name"])]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    stats = sorted(glob.glob(os.path.join(src, "stats", "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    for f in stats[-1:]:
        rows = list(csv.reader(open(f)))
        with open(os.path.join(dst, "kernel_stats.csv"), "w", newline="") as out:
            w = csv.writer(out)
            w.writerow(rows[0])
            for r in rows[1:]:
                w.writerow([short(r[0])] + r[1:])
    fetch = counter_means(os.path.join(src, "pmc_fetch"), "FETCH_SIZE")
    write = counter_means(os.path.join(src, "pmc_write"), "WRITE_SIZE")
    per = {}
    for k in sorted(set(fetch) | set(write)):
        if "mil::" not in k and "fft_rtc" not in k:
            continue
        fk, wk = fetch.get(k, (0.0, 0)), write.get(k, (0.0, 0))
        per[k] = {"launches": max(fk[1], wk[1]), "fetch_kib_raw": round(fk[0], 1), "fetch_kib_x2": round(2 * fk[0], 1),
                  "write_kib": round(wk[0], 1), "hbm_bytes_per_launch": int((2 * fk[0] + wk[0]) * 1024)}
    if per:
        json.dump(per, open(os.path.join(dst, "pmc_per_kernel.json"), "w"), indent=1)
        tfile = os.path.join(os.path.dirname(os.path.abspath(dst)), "pmc_traffic.json")
        traffic = json.load(open(tfile)) if os.path.exists(tfile) else {}
        for backend, pats in OURS.items():
            for pat, key in pats.items():
                for k, v in per.items():
                    if re.search(pat, k):
                        traffic.setdefault(backend, {})[key] = v["hbm_bytes_per_launch"]
        # provenance: bench.py drops `traffic` when a kernel source changed after this measurement
        import subprocess
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        import bench
        traffic["sources_sha256"] = bench.kernel_source_hashes()
        traffic["commit"] = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True,
                                           text=True).stdout.strip() or None
        symbols = traffic.setdefault("symbols", {})
        for backend, pats in OURS.items():
            for pat, key in pats.items():
                for k in per:
                    if re.search(pat, k):
                        symbols[key] = k
        traffic["source"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py at 512x512x256; "
                             "bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950 FETCH_SIZE correction, "
                             "MI355X_MICROARCH.md)")
        json.dump(traffic, open(tfile, "w"), indent=1)
    print("wrote", dst)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
