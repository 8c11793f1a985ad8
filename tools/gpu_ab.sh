cd /root/repo
mkdir -p gpurun_out/ab
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.err
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.load(open(f"gpurun_out/ab/{n}.json"))
    fam=d["extra"]["kernel_families"]
    print(n, d["value"], "it/s", {k.split(" ")[0]: v["ms_per_launch"] for k,v in fam.items()})
except Exception as e:
    print(n, "FAILED", e)
PY
}
timeout 900 python -m pytest tests/test_gpu_fft.py tests/test_gpu_decon.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head
MIL_FX_XROWS=16 timeout 900 python -m pytest tests/test_gpu_fft.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head
run rows8 MIL_FX_XROWS=8
run rows16 MIL_FX_XROWS=16
run no_fft MIL_FX_ABLATE=3
