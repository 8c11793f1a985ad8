#!/usr/bin/env python3
"""Host round trip between dependent cost evaluations: the rocprofv3 kernel trace of tools/reg_gap.py (one whole registration)
-> durations of the cost kernels and the gaps between the end of one and the start of the next.

    python tools/reg_gap_summary.py <rocprof output dir> <stdout log of reg_gap.py>
"""
import csv, glob, os, sys
import numpy as np

src, log = sys.argv[1], sys.argv[2]
files = sorted(glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(files[-1])) if "corr_" in r["Kernel_Name"] and "kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
start = np.array([int(r["Start_Timestamp"]) for r in rows], dtype=np.int64)
end = np.array([int(r["End_Timestamp"]) for r in rows], dtype=np.int64)
dur = (end - start) / 1e3
gap = (start[1:] - end[:-1]) / 1e3
gap = gap[gap < 200]  # the pauses between the stages of the cascade (final resample, re-padding) are not round trips
names = {}
for r in rows:
    n = r["Kernel_Name"].split("(")[0].split("::")[-1]
    names[n] = names.get(n, 0) + 1
print("tools/reg_gap.py under `rocprofv3 --kernel-trace` (bench.py's reg3D workload, affMethod 7, 512x512x256):")
for l in open(log):
    if l.startswith("REG"):
        print("  " + l.strip() + "  (profiler attached)")
print(f"  cost kernels in the trace: {len(rows)}, mean {dur.mean():.1f} us, median {np.median(dur):.1f} us; by kernel: {names}")
print(f"  gap between the end of one cost kernel and the start of the next: median {np.median(gap):.2f} us, mean {gap.mean():.2f} us")
