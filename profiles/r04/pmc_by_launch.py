#!/usr/bin/env python3
"""Counters of the LAST scan of a PMC directory (profiles/r04/pmc.sh) per LAUNCH: tier 0, each lean class and the giants separately
(the per-kernel summary lumps every scan_kernel launch together).  usage: pmc_by_launch.py gpurun_out/r04_pmc_<tag>"""
import collections, csv, glob, sys
d = sys.argv[1]
out = collections.defaultdict(dict)
for f in sorted(glob.glob(d + "/p*/*/*_counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("scan_kernel", "rows_kernel", "giant", "reduce_acc"))]
    red = sorted({int(r["Dispatch_Id"]) for r in rows if "reduce_acc" in r["Kernel_Name"]})
    cut = red[-2] if len(red) > 1 else -1
    for r in rows:
        if int(r["Dispatch_Id"]) <= cut or "reduce" in r["Kernel_Name"]:
            continue
        k = ("giant_kernel" if "giant" in r["Kernel_Name"] else "rows_kernel" if "rows_kernel" in r["Kernel_Name"] else "scan_kernel") + " grid=%s" % r["Grid_Size"]
        out[k][r["Counter_Name"]] = out[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, c in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    g = lambda n: c.get(n, 0.0)
    wc, w = g("SQ_WAVE_CYCLES"), max(g("SQ_WAVES"), 1)
    print("%-30s waves %9d | VALU instr %.3e (%7.0f per wave) | wave-cycles/4 %.3e (%7.0f per wave) | VALU active %.3f of a wave's cycles, lanes %.2f, waiting %.2f | LDS instr per wave %5.0f | FETCH_SIZE KB %.3e"
          % (k, w, g("SQ_INSTS_VALU"), g("SQ_INSTS_VALU") / w, wc, wc / w, g("SQ_ACTIVE_INST_VALU") / max(wc, 1), g("SQ_THREAD_CYCLES_VALU") / 64 / max(g("SQ_ACTIVE_INST_VALU"), 1), g("SQ_WAIT_ANY") / max(wc, 1), g("SQ_INSTS_LDS") / w, g("FETCH_SIZE")))
