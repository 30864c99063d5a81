#!/usr/bin/env python3
"""Copies the summaries of gpurun_out/final (written by profiles/final.sh) into profiles/ under a round tag."""
import collections, csv, glob, json, os, shutil, sys
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
tag = sys.argv[1]
O = "gpurun_out/final"
for f in ("eu", "web", "w0"):
    line = [l for l in open("%s/bench_%s.log" % (O, f)) if l.startswith('{"metric"')][-1]
    open("profiles/%s_%s8g_bench.json" % (tag, f), "w").write(line)
line = [l for l in open(O + "/dist1.log") if l.startswith('{"metric"')][-1]
open("profiles/%s_eu2g_torchrun1_bench.json" % tag, "w").write(line)
shutil.copy(newest(O + "/stats/*/*_kernel_stats.csv"), "profiles/%s_eu8g_kernel_stats.csv" % tag)
out = {}
for d in ("pmc_fetch", "pmc_valu", "pmc_misc"):
    rows = list(csv.DictReader(open(newest("%s/%s/*/*_counter_collection.csv" % (O, d)))))
    rows = [r for r in rows if "rows_kernel" in r["Kernel_Name"] or "decode_kernel" in r["Kernel_Name"]]
    # only the LAST scan of the run (steady state: skip index built, tiers learned): everything dispatched after the
    # second-to-last tier-0 launch (tier 0 = the rows_kernel dispatches within 10 % of the largest grid)
    gmax = max(int(r["Grid_Size"]) for r in rows if "rows_kernel" in r["Kernel_Name"])
    big = sorted({int(r["Dispatch_Id"]) for r in rows if "rows_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 0.9 * gmax})
    cut = big[-2] if len(big) > 1 else -1
    rows = [r for r in rows if int(r["Dispatch_Id"]) > cut]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        kn = r["Kernel_Name"]
        name = ("rows_kernel" if "rows_kernel" in kn else "decode_kernel<slow>") + " lds=%s grid=%s" % (r["LDS_Block_Size"], r["Grid_Size"])
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, cs in agg.items():
        for k, v in cs.items():
            out.setdefault(name, {})[k] = {"launches": len(v), "mean_per_launch": sum(v) / len(v), "sum": sum(v)}
json.dump(out, open("profiles/%s_eu8g_pmc.json" % tag, "w"), indent=1)
b = json.loads(open("profiles/%s_eu8g_bench.json" % tag).read())
arcs = b["config"]["arcs_per_gpu"]
t0 = max((k for k in out if k.startswith("rows_kernel")), key=lambda k: out[k].get("SQ_INSTS_VALU", {}).get("mean_per_launch", 0))
v = out[t0]
print("tier0:", t0)
print("  VALU/arc %.2f  SALU/arc %.2f  active lanes %.3f" % (v["SQ_INSTS_VALU"]["mean_per_launch"] / arcs, v["SQ_INSTS_SALU"]["mean_per_launch"] / arcs,
      v["SQ_THREAD_CYCLES_VALU"]["mean_per_launch"] / 64 / v["SQ_ACTIVE_INST_VALU"]["mean_per_launch"]))
# FETCH_SIZE of the last scan: every decode dispatch after the second-to-last tier-0 launch
fetch = sum(c["FETCH_SIZE"]["sum"] for c in out.values() if "FETCH_SIZE" in c)
print("  FETCH_SIZE (KB, all launches of the last scan) %.4g -> x1024 x2 = %.4g bytes" % (fetch, fetch * 2048))
json.dump({"eu": {"hbm_bytes_per_launch": fetch * 2048,
                  "note": "FETCH_SIZE summed over every decode launch of the last scan of the counter pass, x1024 B, x2 (gfx950 wide-read correction, MI355X_MICROARCH.md HBM section)"}},
          open("profiles/traffic.json", "w"), indent=1)
print("  bench:", b["value"], b["ms_per_step"], b["roofline"]["achieved"], b["roofline"]["frac"], "cpu", b["cpu_baseline"]["value"], b["cpu_baseline"]["cores"])
for k in ("web", "w0"):
    bb = json.loads(open("profiles/%s_%s8g_bench.json" % (tag, k)).read())
    print(" ", k, bb["value"], bb["roofline"]["achieved"], bb["roofline"]["frac"], bb["config"]["bits_per_link"], "cpu", bb["cpu_baseline"]["value"])
