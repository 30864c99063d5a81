#!/usr/bin/env python3
"""Copies the outputs of profiles/r02/final.sh (gpurun_out/r02_final) into profiles/ under the r02 tag and writes profiles/traffic.json
(the PMC-measured HBM traffic of the default bench workload, with the configuration it was measured on: bench.py prints it as
roofline.traffic only for that very configuration)."""
import glob, json, os, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
O = os.path.join(R, "gpurun_out", "r02_final"); P = os.path.join(R, "profiles")
cp = lambda a, b: shutil.copy(os.path.join(O, a), os.path.join(P, b))
for a, b in [("bench_eu15.json", "r02_eu15_bench.json"), ("bench_eu.json", "r02_eu8g_bench.json"), ("bench_web.json", "r02_web8g_bench.json"), ("bench_w0.json", "r02_w08g_bench.json"),
             ("bench_eu_2g3.json", "r02_eu_2g3nodes_bench.json"), ("bench_eu_u64.json", "r02_eu_u64_bench.json"), ("bench_eu15_torchrun1.json", "r02_eu15_torchrun1_bench.json"), ("strong_rehearsal.txt", "r02_strong_rehearsal.txt"),
             ("speedtest.json", "r02_speedtest.json"), ("ktrace.txt", "r02_eu15_scan_timeline.txt"), ("pmc_summary.txt", "r02_eu15_pmc_summary.txt"), ("pmc_summary.json", "r02_eu15_pmc.json"),
             ("prof.txt", "r02_section_timers.txt"), ("ldspad.txt", "r02_occupancy_ldspad.txt")]:
    if os.path.exists(os.path.join(O, a)):
        cp(a, b)
for f in glob.glob(O + "/kt/*/*_kernel_stats.csv"):
    shutil.copy(f, os.path.join(P, "r02_eu15_kernel_stats.csv"))
for f in glob.glob(O + "/kt/*/*_kernel_trace.csv"):
    shutil.copy(f, os.path.join(P, "r02_eu15_kernel_trace.csv"))          # the raw per-dispatch trace: profiles/union.py and r02/ktrace_summary.py reproduce from it
pm = json.load(open(os.path.join(P, "r02_eu15_pmc.json")))
b = json.load(open(os.path.join(P, "r02_eu15_bench.json")))
fetch_kb = sum(k.get("FETCH_SIZE", 0.0) for k in pm["kernels"].values())
sha = subprocess.check_output(["git", "-C", R, "rev-parse", "--short", "HEAD"]).decode().strip()
json.dump({"runs": [{"shape": b["config"]["shape"], "tiles": b["config"]["tiles"], "base_nodes": b["config"]["base_nodes"], "n_gpus": 1, "scaling": "weak",
                     "hbm_bytes_per_launch": fetch_kb * 2048.0,
                     "source": "profiles/r02_eu15_pmc.json: FETCH_SIZE (KB) summed over every kernel of the last scan of `rocprofv3 --pmc FETCH_SIZE -- python bench.py --shape eu15 --steps 2 --warmup 0` (profiles/r02/final.sh, tree at %s) x 1024 B x 2 (gfx950 counts wide reads at half, MI355X_MICROARCH.md HBM section)" % sha}]},
          open(os.path.join(P, "traffic.json"), "w"), indent=1)
print("traffic: %.4g bytes per scan for %.4g algorithmic + %.4g index bytes" % (fetch_kb * 2048.0, b["roofline"]["algorithmic_bytes_per_launch"], b["roofline"]["index_bytes_per_launch"]))
