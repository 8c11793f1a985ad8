# xy kernel efficiency per NH = nx / 2: boxes 128 x 512 x nx, nx = 192 .. 1024, under rocprofv3; GB/s = (2 C'' + R) / t for the ratio
# kernel (MODE 2) and (2 C'' + 2 R) / t for the update kernel (MODE 3)
B=""
for nx in 192 256 320 384 448 512 576 640 704 768 832 896 960 1024; do B="$B;128,512,$nx"; done
B=${B#;}
MIL_FX_ANY=${MIL_FX_ANY:-1} bash tools/prof_cmd.sh xysweep python tools/xy_any_probe.py "--boxes=$B" --iters=30 > gpurun_out/xysweep_raw.log 2>&1
python - <<'PY'
import re
rows = {}
for line in open("gpurun_out/xysweep_raw.log"):
    m = re.match(r"xy(2?)_kernel<(\d), (?:(\d+), )?(\d+), (true|false)>.*avg\s+([\d.]+) us", line)
    if m:
        mode, nh, us = int(m.group(2)), int(m.group(4)), float(m.group(6))
        rows.setdefault(nh, {})[mode] = us
        rows[nh]["form"] = "two threads per column" if m.group(1) else "one thread per column"
for nh in sorted(rows):
    nz, ny = 128, 512
    C = nz * ny * (nh + 16) * 8
    R = nz * ny * 2 * nh * 4
    r = rows[nh]
    print(f"NH {nh:4d} (M = {nh // 16:2d}, {r.get('form', '')}): ratio {r.get(2, 0):7.1f} us {(2 * C + R) / r[2] / 1e3 if 2 in r else 0:7.0f} GB/s   update {r.get(3, 0):7.1f} us {(2 * C + 2 * R) / r[3] / 1e3 if 3 in r else 0:7.0f} GB/s")
PY
grep "it/s" gpurun_out/xysweep_raw.log
