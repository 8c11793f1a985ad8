# copy the summaries of gpurun_out/r06 (tools/gpu_profile_run_r06.sh) into the tracked profiles/r06 and refresh profiles/pmc_traffic.json
o=gpurun_out/r06; p=profiles/r06
cp $o/summary/kernel_stats.csv $o/summary/pmc_per_kernel.json $o/summary/reg_cost_kernel.json $p/
cp $o/bench.json $o/bench_steps20.json $o/bench_rocfft_backend.json $o/bench_2ranks_self_launched_gloo.json $o/bench_1rank_rccl_forced.json $o/bench_sizes.json $o/reg_host_gap.txt $o/resample_ab.txt $o/gpu_tests.log $p/
cp $o/twopass/summary.json $p/twopass_pmc.json
cp $o/aux/summary.json $p/aux_pmc.json
cp $(ls -S $o/aux/trace/runc/*_kernel_stats.csv | head -1) $p/aux_kernel_stats.csv
python - <<'PY'
import json, subprocess, sys
sys.path.insert(0, ".")
import bench
d = json.load(open("gpurun_out/r06/pmc_traffic.json"))
cur = bench.kernel_source_hashes()
assert all(d["sources_sha256"].get(k) == v for k, v in cur.items()), "kernel sources changed since the PMC passes"
d["commit"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
json.dump(d, open("profiles/pmc_traffic.json", "w"), indent=1)
b = json.load(open("profiles/r06/bench.json"))
print(b["value"], b["roofline"]["frac"], b["roofline"]["traffic"], b["extra"]["errors"])
PY
