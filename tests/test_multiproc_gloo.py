"""CPU, world_size 2 over gloo: the N>1 paths of bench.py.  Both go through webgraph-big_amd/shard.py, the very helpers
bench.py calls (shard.sharded_scan / allreduce_scan / allreduce_max):
  * weak scaling: node-range shards with a node-id base; the reduced {arcs, chk} equal those of the whole (tiled) graph;
  * strong scaling: ONE graph, rank r scans [bounds[r], bounds[r+1]) of the arc-balanced split; the reduced pair equals the
    one-piece scan (BASELINE config 5).
The per-shard scans run on the CPU oracle where there is no GPU (this container); where there is one, the strong-scaling half
hands the helper the HIP handle's scan (two ranks sharing the device), as bench.py does with RCCL in place of gloo."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, seed, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import webgraph_big_amd as W
    import tooling as T
    from webgraph_big_amd import shard as S
    from oracle import bvg_oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    st = T.synth_store(n, seed=seed, chunk_nodes=2048, threads=2)       # every rank holds the same shard bytes (weak scaling)
    og = O.Graph.from_memory(O.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)
    # weak: shard `rank` of the world-times larger graph (bench.py --scaling weak)
    _, arcs, chk = S.sharded_scan(lambda lo, hi: og.scan(lo, hi, node_base=rank * n), [0] * rank + [0, n] + [n] * world, rank)
    tmax = S.allreduce_max(float(rank))
    # strong: one graph, arc-balanced node ranges (bench.py --scaling strong)
    deg, _ = og.decode_range(0, n)
    bounds = S.bounds_by_arcs(deg, world)
    scan = lambda lo, hi: og.scan(lo, hi)
    import torch
    if torch.cuda.device_count() > 0:                                    # a GPU box: the product handle does the scanning, gloo the reduction
        hg = W.BVGraph.from_memory(st.params, st.graph, st.offsets, device=0)
        assert hg.shard_bounds(world, W.BALANCE_ARCS).tolist() == bounds.tolist()
        scan = lambda lo, hi: hg.scan(lo, hi)
    own, sarcs, schk = S.sharded_scan(scan, bounds, rank)
    if rank == 0:
        q.put((arcs, chk, tmax, sarcs, schk, bounds.tolist(), own["arcs"]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_scan_matches_whole_graph(W, tools, oracle):
    import torch.multiprocessing as mp
    n, seed, world = 6000, 21, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, seed, q)) for r in range(world)]
    for p in procs: p.start()
    arcs, chk, tmax, sarcs, schk, bounds, own_arcs = q.get(timeout=120)
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    assert tmax == 1.0
    # the whole graph: `world` translated copies of the shard, stored explicitly and scanned in one piece
    off, adj = tools.synth_adjacency(n, seed=seed, chunk_nodes=2048)
    deg = np.diff(off).astype(np.int64)
    big_adj = np.concatenate([adj + j * n for j in range(world)])
    big_off = np.concatenate([[0], np.cumsum(np.tile(deg, world))]).astype(np.uint64)
    st = tools.store((big_off, big_adj), W.default_params(), chunk_nodes=2048)
    og = oracle.Graph.from_memory(oracle.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)
    whole = og.scan()
    assert (arcs, chk) == (whole["arcs"], whole["chk"])
    # strong scaling: the two shards of ONE graph add up to its one-piece scan, and the split is arc-balanced
    st1 = tools.synth_store(n, seed=seed, chunk_nodes=2048, threads=2)
    og1 = oracle.Graph.from_memory(oracle.Params(**st1.params.as_dict()), st1.graph.tobytes(), st1.offsets)
    one = og1.scan()
    assert (sarcs, schk) == (one["arcs"], one["chk"])
    assert bounds[0] == 0 and bounds[-1] == n and 0 < bounds[1] < n
    assert abs(own_arcs - one["arcs"] / 2) <= 0.02 * one["arcs"] + 3000


def test_bounds_by_arcs_rule(W):
    from webgraph_big_amd import shard as S
    deg = np.array([3, 0, 0, 5, 1, 1, 0, 2], dtype=np.int32)            # cum 0 3 3 3 8 9 10 10 12; per = ceil(12/3) = 4
    assert S.bounds_by_arcs(deg, 3).tolist() == [0, 4, 4, 8]             # first node whose exclusive prefix reaches 4 / 8 (node 3 alone holds 5 arcs)
    assert S.bounds_by_arcs(deg, 1).tolist() == [0, 8]
    assert S.bounds_by_arcs(np.zeros(5, np.int32), 2).tolist() == [0, 5, 5]


def test_split_nodes_matches_reference_rule(W):
    from webgraph_big_amd import shard as S
    assert S.split_nodes(10, 3) == [(0, 4), (4, 8), (8, 10)]
    assert S.split_nodes(3, 5) == [(0, 1), (1, 2), (2, 3), (3, 3), (3, 3)]     # extras are empty (NodeIterator.EMPTY)
    assert S.split_nodes(0, 2) == [(0, 0), (0, 0)]
    assert S.i64_to_u64(S.u64_to_i64((1 << 64) - 5)) == (1 << 64) - 5
