#!/usr/bin/env python3
"""Adds (or replaces) the profiles/traffic.json entry of a bench shape from a finished PMC run (profiles/r06/pmc.sh <tag> --shape S ...): HBM bytes per launch
(FETCH_SIZE x 1024 x 2), VALU / SALU / LDS wave-instructions per arc, active lanes -- stamped with the hash of the kernel sources they were measured on
(bench.py kernel_source_id): bench.py quotes the entry only while webgraph-big_amd/csrc is byte-identical.
    python3 profiles/r06/record.py <tag> <shape> <tiles> <base_nodes> <summary file under profiles/>"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
tag, shape, tiles, base_nodes, summ = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
s = json.load(open(os.path.join(ROOT, "gpurun_out", "r06_pmc_%s" % tag, "summary.json")))
arcs = float(s["arcs"]); ks = s["kernels"]
tot = lambda n: sum(k.get(n, 0.0) for k in ks.values())
dom = max(ks.values(), key=lambda k: k.get("SQ_WAVE_CYCLES", 0.0))
dom_name = max(ks, key=lambda k: ks[k].get("SQ_WAVE_CYCLES", 0.0))
e = {"rev": bench.KERNEL_REV, "src_id": bench.kernel_source_id(), "shape": shape, "tiles": tiles, "base_nodes": base_nodes, "n_gpus": 1, "scaling": "weak",
     "hbm_bytes_per_launch": tot("FETCH_SIZE") * 1024.0 * 2.0 if tot("FETCH_SIZE") else None,
     "valu_per_arc": tot("SQ_INSTS_VALU") / arcs, "salu_per_arc": tot("SQ_INSTS_SALU") / arcs, "lds_per_arc": tot("SQ_INSTS_LDS") / arcs,
     "active_lanes": dom["SQ_THREAD_CYCLES_VALU"] / 64.0 / dom["SQ_ACTIVE_INST_VALU"],
     "source": "profiles/%s: FETCH_SIZE (KB) summed over every kernel of the last scan of `rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 bench.py --shape %s --steps 2 --warmup 2 --no-cpu-baseline --no-verify --no-index-leg` (profiles/r06/pmc.sh) x 1024 B x 2 (gfx950 counts wide reads at half, MI355X_MICROARCH.md HBM section); kernel sources %s" % (summ, shape, bench.kernel_source_id()),
     "valu_source": "profiles/%s: SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_INSTS_LDS summed over every kernel of the last scan (their own --pmc pass of the same command) / arcs of the scan; active lanes = SQ_THREAD_CYCLES_VALU / 64 / SQ_ACTIVE_INST_VALU of the dominant kernel (%s)" % (summ, dom_name)}
p = os.path.join(ROOT, "profiles", "traffic.json")
rec = json.load(open(p))
rec["runs"] = [r for r in rec["runs"] if not (r.get("rev") == e["rev"] and r.get("shape") == shape and r.get("tiles") == tiles and r.get("base_nodes") == base_nodes)] + [e]
json.dump(rec, open(p, "w"), indent=1)
print(json.dumps(e, indent=1))
