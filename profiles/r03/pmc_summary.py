#!/usr/bin/env python3
"""Sums the counters of the LAST scan of each PMC pass per kernel (name + LDS size), prints ratios per arc and per wave-cycle,
and writes <dir>/summary.json.  usage: pmc_summary.py <dir with p1..pN from profiles/r02/pmc.sh>"""
import collections, csv, glob, json, os, re, sys
d = sys.argv[1]
out = collections.defaultdict(dict)
arcs = None
for log in sorted(glob.glob(d + "/p*.log")):
    for l in open(log):
        if l.startswith('{"metric"'):
            arcs = json.loads(l)["config"]["arcs_per_gpu"]
def short(kn):
    m = re.search(r"(scan_kernel|giant_kernel|rows_wg_kernel|rows_kernel|decode_kernel|reduce_acc_kernel|flat_kernel)(<[^>]*>)?", kn)
    return (m.group(1) + (m.group(2) or "")) if m else None
for f in sorted(glob.glob(d + "/p*/*/*_counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(f)) if short(r["Kernel_Name"])]
    if not rows:
        continue
    # the last scan = every dispatch after the last reduce_acc but one
    red = sorted({int(r["Dispatch_Id"]) for r in rows if "reduce_acc" in r["Kernel_Name"]})
    cut = red[-2] if len(red) > 1 else -1
    for r in rows:
        if int(r["Dispatch_Id"]) <= cut:
            continue
        k = "%s lds=%s" % (short(r["Kernel_Name"]), r["LDS_Block_Size"])
        out[k][r["Counter_Name"]] = out[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        out[k].setdefault("_grid", 0); out[k].setdefault("_wg", int(r.get("Workgroup_Size", 64) or 64))
        if r["Counter_Name"] in ("SQ_WAVES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE", "FETCH_SIZE"):
            pass
json.dump({"arcs": arcs, "kernels": out}, open(d + "/summary.json", "w"), indent=1)
print("arcs per scan:", arcs)
tot = collections.Counter()
for k, c in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if not wc:
        continue
    g = lambda n: c.get(n, 0.0)
    print("%-60s waves %.3g" % (k, g("SQ_WAVES")))
    print("    per arc (all arcs of the scan): VALU %.3f SALU %.3f LDS %.3f SMEM %.3f VMEM_RD %.3f BRANCH %.3f" % tuple(g(n) / arcs for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_BRANCH")))
    print("    per wave-cycle: active any %.3f (VALU %.3f SCA %.3f LDS %.3f VMEM %.3f) wait_any %.3f wait_inst_any %.3f wait_inst_lds %.3f | active lanes %.3f | LDS conflict cycles / LDS instr %.2f" % (
        g("SQ_ACTIVE_INST_ANY") / wc, g("SQ_ACTIVE_INST_VALU") / wc, g("SQ_ACTIVE_INST_SCA") / wc, g("SQ_ACTIVE_INST_LDS") / wc, g("SQ_ACTIVE_INST_VMEM") / wc,
        g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_WAIT_INST_LDS") / wc, g("SQ_THREAD_CYCLES_VALU") / 64 / max(g("SQ_ACTIVE_INST_VALU"), 1), g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_INSTS_LDS"), 1)))
    print("    wave-cycles (quad) %.4g  busy cycles %.4g  instr per wave-quad-cycle %.3f  FETCH_SIZE KB %.4g" % (wc, g("SQ_BUSY_CYCLES"), (g("SQ_INSTS_VALU") + g("SQ_INSTS_SALU") + g("SQ_INSTS_LDS") + g("SQ_INSTS_SMEM") + g("SQ_INSTS_VMEM_RD")) / wc, g("FETCH_SIZE")))
    for n in c:
        if not n.startswith("_"):
            tot[n] += c[n]
print("all kernels of the scan: VALU/arc %.3f SALU/arc %.3f  FETCH_SIZE %.4g KB -> x1024 x2 = %.4g bytes" % (tot["SQ_INSTS_VALU"] / arcs, tot["SQ_INSTS_SALU"] / arcs, tot["FETCH_SIZE"], tot["FETCH_SIZE"] * 2048))
