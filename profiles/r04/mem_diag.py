"""Where the resident bytes of the default workload go: free HBM after every phase (BVG_DEBUG=1 prints the index / giant figures)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import webgraph_big_amd as W
import tooling as T
tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 128
f0 = torch.cuda.mem_get_info(0)[0]
def used(tag):
    torch.cuda.synchronize()
    print("%-34s %8.2f GB in use" % (tag, (f0 - torch.cuda.mem_get_info(0)[0]) / 1e9), flush=True)
used("start (torch context)")
sts = [T.synth_store(1 << 20, seed=sd, params=W.default_params(), synth=T.eu_like(**kw), threads=16) for sd, kw in bench.MIX]
bases = [W.BVGraph.from_memory(st.params, st.graph, st.offsets, device=0) for st in sts]
used("8 bases")
g = W.mosaic(bases, tiles)
used("mosaic (stream + packed offsets)")
print("stream %.2f GB, nodes %.2f G" % (sum(len(s.graph) for s in sts) * tiles / 1e9, g.num_nodes() / 1e9))
g.build_index(); used("after build_index")
r = g.scan(); used("after scan 1")
r = g.scan(); used("after scan 2")
print({k: r[k] for k in ("index_bytes", "index_entries", "lean_blocks", "slow_blocks")})
h = g.copy(); h.set_tuning(no_index=True); h.scan(); used("after a no-index flyweight's scan"); h.close(); used("flyweight closed")
