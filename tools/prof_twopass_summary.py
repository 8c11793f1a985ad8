"""Per-kernel means of the rocprofv3 passes written by tools/prof_twopass.sh -> <dir>/summary.json (+ stdout)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
out = defaultdict(dict)


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|mil::fx::", "", n)
    return n[:60]


for f in glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        a = acc[(short(r["Kernel_Name"]), r["Counter_Name"])]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    for (k, c), (s, n) in acc.items():
        out[k][c] = s / n
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        out[short(r["Name"])]["avg_us"] = float(r["AverageNs"]) / 1e3
        out[short(r["Name"])]["calls"] = int(r["Calls"])
res = {}
for k, v in out.items():
    if "avg_us" not in v or v.get("calls", 0) < 4:
        continue
    d = dict(v)
    if "FETCH_SIZE" in d:
        d["fetch_MB_x2"] = d.pop("FETCH_SIZE") * 2 / 1024
    if "WRITE_SIZE" in d:
        d["write_MB"] = d.pop("WRITE_SIZE") / 1024
    if "SQ_BUSY_CYCLES" in d and "SQ_ACTIVE_INST_VALU" in d:
        d["valu_busy_frac_of_wave_cycles"] = d["SQ_ACTIVE_INST_VALU"] / max(d.get("SQ_WAVE_CYCLES", 1), 1)
    if "SQ_WAVE_CYCLES" in d and d.get("SQ_WAVES"):
        w = d["SQ_WAVES"]
        d["per_wave_kcycles"] = {c[3:].lower(): round(d[c] * 4 / w / 1e3, 2) for c in
                                 ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                                  "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC",
                                  "SQ_WAIT_INST_LDS") if c in d}
        d["per_wave_insts"] = {c[9:].lower(): round(d[c] / w, 1) for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_SMEM",
                                                                         "SQ_INSTS_BRANCH") if c in d}
    res[k] = {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in d.items()}
json.dump(res, open(os.path.join(root, "summary.json"), "w"), indent=1)
for k, v in res.items():
    if 'yz_conv' in k or 'xy_kernel' in k or 'x_kernel' in k or 'col_kernel' in k:
        print(k, json.dumps({kk: vv for kk, vv in v.items() if not kk.startswith('SQ_')}))
