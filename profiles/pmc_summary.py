#!/usr/bin/env python3
"""Summarises the counter CSVs written by profiles/pmc.sh: per-launch means for the decode kernels."""
import collections, csv, glob, json, sys
tag = sys.argv[1]; arcs = float(sys.argv[2]) if len(sys.argv) > 2 else None
out = {}
for f in sorted(glob.glob("gpurun_out/%s/*/*/*_counter_collection.csv" % tag)):
    agg = collections.defaultdict(list); kt = {}
    for r in csv.DictReader(open(f)):
        if "stream_kernel" in r["Kernel_Name"] or ("decode_kernel" in r["Kernel_Name"] and ", false>" in r["Kernel_Name"] and "true, false>" not in r["Kernel_Name"]):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            kt.setdefault((r["Counter_Name"]), (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Kernel_Name"][:60]))
    for k, v in agg.items():
        out[k] = sum(v) / len(v)
    if kt: out["_kernel"] = list(kt.values())[0]
if arcs:
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
        if k in out: out[k + "_per_arc"] = out[k] / arcs
if "SQ_THREAD_CYCLES_VALU" in out and "SQ_ACTIVE_INST_VALU" in out:
    out["active_lane_frac"] = out["SQ_THREAD_CYCLES_VALU"] / (64.0 * out["SQ_ACTIVE_INST_VALU"])
print(json.dumps(out, indent=1))
