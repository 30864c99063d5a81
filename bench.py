#!/usr/bin/env python3
"""bench.py — decoded edges/s of a full sequential BVGraph successor scan on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): decoded edges/s of a full sequential successor scan, plus the achieved
fraction of the HBM-read roofline (algorithmic bytes = size of the .graph stream).

Workload (config.workload), default `--shape eu15`: the eu-2015 configuration's synthetic STAND-IN at eu-2015's scale -- a MOSAIC
of 8 different copy-model web graphs (2^20 nodes each; different seeds, outdegree mixes, copy and interval parameters; default BV
parameters: window 7, maxRef 3, minInterval 4, zeta_3), each generated and compressed on the host by the repo's own encoder, their
streams concatenated on the device and the cycle repeated 128 times (bvg_mosaic; BV records are translation invariant): 2^30 = 1.07 G
nodes, ~92 G arcs (eu-2015: 1.07 G nodes, 91.8 G arcs), a ~30 GB .graph stream resident in HBM, two orders of magnitude beyond the
256 MiB Infinity Cache.  No LAW dataset can reach the GPU box (no network), hence "stand-in".  `--shape eu15mono` is round 2's form
of it (ONE 2^21-node tile repeated 512 times), `eu` round 1's smaller graph (8 GiB, 26 G arcs), `web` / `w0` BASELINE configs 3 / 2.
`--shape cnr` is the one REAL web graph in the tree: the reference's own fixture cnr-2000 (tests/golden/, byte-identical to
slow/it/unimi/dsi/big/webgraph/cnr-2000.* of the reference; 325 557 nodes, 3.2 M arcs, W=7 maxRef=3 minInterval=3), its stream repeated
on the device (bvg_mosaic of one base) to 8 GiB -- real LAW structure (24 % empty nodes, chains of depth 3, 9.9 arcs per node) at a size
that defeats the caches.
`--basename PATH` benchmarks a real BVGraph instead (PATH.properties / .graph / .offsets, as test/SpeedTest.java:117-146 takes it):
loaded through bvg_open, sample node ranges gated against the CPU oracle, the CPU baseline timed on a prefix of its nodes.

A "step" is one full scan of every node of the graph: successors are decoded, counted and checksummed
on chip.  N GPUs (one process each, launched by torch.distributed.run):
  --scaling strong (default for N > 1; BASELINE config 5): every rank holds a replica of the ONE graph
      and scans shard `rank` of the arc-balanced N-way node-range split (bvg_shard_bounds /
      ImmutableGraph.splitNodeIterators); the step ends with the path's only collective, one RCCL
      all-reduce of {arcs, chk}, and the reduced pair must equal the one-piece scan of the whole graph;
  --scaling weak: each rank scans a whole graph of its own (node ids shifted by rank * nodes).
Both go through webgraph-big_amd/shard.py, the helpers tests/test_multiproc_gloo.py runs over gloo.
Protocol mirrors the reference's SpeedTest (3 warm-up + 10 timed scans).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)

SHAPES = {
    # name: (synth kwargs for tools.eu_like / web_like, params kwargs, default tiles of a 2^21-node base, description)
    "eu15": ("mix", {}, {}, 128, "eu-2015 stand-in at eu-2015 scale: mosaic of 8 synthetic copy-model graphs, W=7 maxRef=3 minInterval=4 zeta3"),
    "eu15mono": ("eu", dict(mean_deg=127.5), {}, 512, "eu-2015 stand-in at eu-2015 scale (round 2's form: one tile repeated): synthetic copy model, W=7 maxRef=3 minInterval=4 zeta3"),
    "eu": ("eu", {}, {}, 0, "eu-2015-shaped synthetic (copy model, W=7 maxRef=3 minInterval=4 zeta3)"),
    "web": ("web", {}, {}, 0, "cnr/uk-shaped synthetic (copy model, W=7 maxRef=3 minInterval=4 zeta3)"),
    "uk": ("web", dict(mean_deg=44.0), {}, 0, "uk-2007-05-shaped synthetic at its density (35 arcs per node; copy model, W=7 maxRef=3 minInterval=4 zeta3)"),
    "w0": ("web", {}, dict(window_size=0, max_ref_count=0, min_interval_length=0), 0, "uk-2007-05 re-store stand-in (window=0 maxRef=0, zeta3 residuals only)"),
    "cnr": ("golden", {}, {}, 0, "cnr-2000 (LAW; the reference's fixture, W=7 maxRef=3 minInterval=3 zeta3) tiled on the device"),
}
GOLDEN = os.path.join(ROOT, "tests", "golden", "cnr-2000")
KERNEL_REV = "r06"      # profiles/traffic.json entries measured on other kernels are not quoted ...
VALU_PEAK_WINSTR_PER_S = 1024 / 1.83e-9   # 256 CUs x 4 SIMDs, one wave-instruction per 1.83 ns each (profiles/r03_valu_rates.txt, measured on the card)
VALU_PEAK_FULL_RATE = 1024 / 1.07e-9      # ... per 1.07 ns for the full-rate operations (v_add / sub / and / or / xor / lshr / mov: profiles/r04_valu_rates2.txt)


def kernel_source_id():
    """... and an entry counts only for the kernel SOURCES it was measured on: a hash over webgraph-big_amd/csrc (what libbvgraph_hip.so is built from).  A PMC
    pass on file for an earlier tree never decorates a later kernel: `traffic` and `roofline_valu` are null until the pass is repeated (profiles/r05/pmc.sh + record.py)."""
    import glob
    import hashlib
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "webgraph-big_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
VALU_FLOOR_PER_ARC = 30.0 / 64.0          # ~30 lane-operations of decode + checksum per arc on 64 lanes
# the 8 bases of the eu15 mosaic: (seed, eu_like overrides).  Mean outdegree ~86 over the cycle (eu-2015: 85.7); the tiles differ in
# density, in how much they copy and in how long their copy blocks, intervals and residual lists are.
MIX = [(0, dict(mean_deg=127.5)),
       (1, dict(mean_deg=70.0, p_copy=0.80, keep_run=15.0, extra_mean=6.0)),
       (2, dict(mean_deg=190.0, p_copy=0.92, keep_run=35.0, extra_mean=3.0, p_interval=0.6)),
       (3, dict(mean_deg=100.0, p_copy=0.85, keep_run=20.0, skip_run=3.0, local_gap=12.0)),
       (4, dict(mean_deg=150.0, p_copy=0.90, keep_run=30.0, interval_len=35.0, p_interval=0.4)),
       (5, dict(mean_deg=85.0, p_copy=0.75, keep_run=12.0, extra_mean=8.0, p_far=0.06)),
       (6, dict(mean_deg=165.0, p_copy=0.88, keep_run=25.0, tail_alpha=2.3)),
       (7, dict(mean_deg=130.0, p_copy=0.93, keep_run=40.0, skip_run=1.5, extra_mean=2.5, p_interval=0.3))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--base-nodes", type=int, default=0, help="nodes of each generated base graph (default: 2^20 for the eu15 mosaic, 2^21 otherwise)")
    ap.add_argument("--basename", default=None, help="benchmark this BVGraph (basename.properties/.graph/.offsets) instead of a synthetic workload")
    ap.add_argument("--target-gib", type=float, default=0.0, help="size of the tiled .graph stream per GPU (0 = the shape's default: 512 tiles for eu15, 8 GiB otherwise)")
    ap.add_argument("--shape", default="eu15", choices=sorted(SHAPES))
    ap.add_argument("--tiles", type=int, default=0, help="explicit number of tiles of the base graph (overrides --target-gib)")
    ap.add_argument("--allow-wide", action="store_true", help="let the tiled graph pass 2^31 nodes (still the 32-bit successor kernels: they hold every id below 2^32) and, beyond 2^32 - 256 nodes, run the 64-bit kernels (dtype u64)")
    ap.add_argument("--scaling", default=None, choices=["strong", "weak"], help="N > 1: strong = shards of ONE graph (default), weak = one graph per rank")
    ap.add_argument("--balance", default="arcs", choices=["arcs", "bits", "nodes"])
    ap.add_argument("--block-bits", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-gib", type=float, default=1.0, help="size of the stream the CPU baseline scans")
    ap.add_argument("--no-verify", action="store_true", help="skip the per-tile checksum gate against the CPU oracle")
    ap.add_argument("--no-real-leg", action="store_true", help="skip the second, untimed-for-`value` workload behind the timed region: the REAL web graph cnr-2000 tiled to 4 GiB (`real_graph` in the line)")
    ap.add_argument("--no-index-leg", action="store_true", help="skip the three index-less scans behind the timed region (value_no_index)")
    ap.add_argument("--no-wide-leg", action="store_true", help="skip the three scans through the 64-bit-id path behind the timed region (value_wide)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="collective backend (nccl = RCCL; gloo only to rehearse N > 1 ranks on a one-GPU box)")
    ap.add_argument("--echo-ranks", action="store_true", help="plumbing test: every rank prints 'rank r of w' and exits before touching the GPU")
    ap.add_argument("--verify-whole", action="store_true", help="strong scaling: rank 0 also scans the whole graph in one piece (untimed) and compares it with the reduced {arcs, chk}")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ       # under torch.distributed.run, also with one rank
    if not launched and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as a CHILD (nothing in this process has
        # touched the GPU yet, and it never will: it only relays the ranks' output and exit code)
        sys.exit(self_launch(args.gpus))
    if launched and world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    scaling = args.scaling or ("strong" if world > 1 else "weak")
    if args.echo_ranks:
        os.write(1, b"rank %d of %d\n" % (rank, world))                   # one write: the ranks share the pipe
        return

    import numpy as np
    import torch
    import webgraph_big_amd as W
    import tooling as T
    from webgraph_big_amd import shard as S

    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if launched:                                                         # the process group exists whenever a launcher set the env
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    dev = local_rank
    cuda = torch.device("cuda", local_rank) if args.backend == "nccl" else None   # where the collective's 16 bytes live

    # ---- input: a real graph from disk, or synthetic bases generated + compressed on the host, uploaded, tiled on the device ----
    t0 = time.time()
    # host threads of THIS rank: the ranks of one node share its cores (8 ranks generating with every core each would oversubscribe the
    # host 8x); with a launcher the synthetic bases are generated ONCE, by rank 0, and handed to the others through /dev/shm
    ncpu = effective_cpus(os.cpu_count() or 1)
    threads = max(1, min(ncpu // max(world, 1), 64))
    kind, skw, pkw, tiles_default, wl = SHAPES[args.shape]
    free0 = torch.cuda.mem_get_info(dev)[0]
    sts, bases, copies = [], [], 1
    gen_s = upload_s = tile_s = 0.0
    if args.basename:
        wl = "BVGraph %s" % os.path.basename(args.basename)
        g = W.BVGraph.load(args.basename, device=dev)
        torch.cuda.synchronize()
        upload_s = time.time() - t0
        n_graph = g.num_nodes()
        args.base_nodes = n_graph
    else:
        if kind == "golden":
            sts = [golden_store(GOLDEN)]                                  # the reference's fixture, read from tests/golden/ (it travels with the repo)
            args.base_nodes = int(sts[0].params.nodes)
        else:
            if not args.base_nodes:
                args.base_nodes = (1 << 20) if kind == "mix" else (1 << 21)
            params = W.default_params(**pkw)

            def generate(nthreads):
                if kind == "mix":
                    return [T.synth_store(args.base_nodes, seed=sd, params=params, synth=T.eu_like(**kw), threads=nthreads) for sd, kw in MIX]
                return [T.synth_store(args.base_nodes, seed=0, params=params, synth=T.eu_like(**skw) if kind == "eu" else T.web_like(**skw), threads=nthreads)]
            if dist is not None and world > 1:
                # The bases are generated once PER NODE, by its local rank 0 (same seeds: identical on every node), and handed to the node's other ranks through a
                # private directory (mode 0700, a random name that rank 0 broadcasts: nothing predictable in a world-writable place), removed whatever happens.
                import secrets
                tok = torch.tensor([secrets.randbits(62) if rank == 0 else 0], dtype=torch.int64, device=cuda if cuda is not None else "cpu")
                dist.broadcast(tok, src=0)                                # (a plain tensor collective: the same call on RCCL and on gloo)
                sdir = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", "bvg_bench_%016x" % int(tok.item()))
                share = os.path.join(sdir, "b")
                leader = int(os.environ.get("LOCAL_RANK", rank)) == 0
                try:
                    lead_err = None
                    if leader:
                        try:
                            os.mkdir(sdir, 0o700)
                            sts = generate(max(1, min(ncpu, 64)))         # the node's other ranks wait for the status below: its leader may use every core
                            save_stores(share, sts)
                        except Exception as ex:                           # (the other ranks must not sit in a barrier until the collective times out: they learn of it and raise too)
                            lead_err = ex
                    bad = torch.tensor([1 if lead_err is not None else 0], dtype=torch.int64, device=cuda if cuda is not None else "cpu")
                    dist.all_reduce(bad, op=dist.ReduceOp.MAX)            # doubles as the barrier behind the leader's work
                    if int(bad.item()):
                        raise RuntimeError("workload generation failed on a node's leader rank: %r" % (lead_err,))
                    if not leader:
                        sts = load_stores(share)
                    dist.barrier()
                finally:
                    if leader:
                        remove_stores(share)
                        try:
                            os.rmdir(sdir)
                        except OSError:
                            pass
            else:
                sts = generate(threads)
        gen_s = time.time() - t0
        cycle_bytes = sum(len(st.graph) for st in sts)
        cycle_nodes = args.base_nodes * len(sts)
        if args.target_gib > 0 or not tiles_default:
            copies = max(1, int((args.target_gib or 8.0) * (1 << 30) / max(cycle_bytes, 1)))
        else:
            copies = max(1, tiles_default * ((1 << 20) if kind == "mix" else (1 << 21)) // args.base_nodes)
        if args.tiles:
            copies = args.tiles
        if not args.allow_wide:
            copies = max(1, min(copies, ((1 << 31) - 1) // cycle_nodes))        # stay on the 32-bit successor kernels
        t0 = time.time()
        bases = [W.BVGraph.from_memory(st.params, st.graph, st.offsets, device=dev) for st in sts]
        torch.cuda.synchronize()
        upload_s = time.time() - t0
        t0 = time.time()
        g = W.mosaic(bases, copies) if (copies > 1 or len(bases) > 1) else bases[0]
        torch.cuda.synchronize()
        tile_s = time.time() - t0
        n_graph = g.num_nodes()
    if args.block_bits:
        g.set_tuning(block_bits=args.block_bits)
    arcs_graph = sum(st.stats["arcs"] for st in sts) * copies if sts else g.num_arcs()
    bal = {"arcs": W.BALANCE_ARCS, "bits": W.BALANCE_BITS, "nodes": W.BALANCE_NODES}[args.balance]
    if scaling == "strong":
        bounds = g.shard_bounds(world, bal)                              # shard `rank` of the ONE graph
        my_rank = rank
    else:
        g.set_node_base(rank * n_graph)                                  # a graph of its own per rank, ids shifted
        bounds = np.array([0, n_graph], dtype=np.int64)
        my_rank = 0
    lo, hi = int(bounds[my_rank]), int(bounds[my_rank + 1])

    def step():
        # the hot path of this rank + the path's only collective (16 bytes over RCCL when a process group exists)
        return S.sharded_scan(lambda a, b: g.scan(a, b), bounds, my_rank, device=cuda if dist is not None else None, reduce=dist is not None)

    t0 = time.time()
    r, tot_arcs, tot_chk = step()                                        # first scan: builds the block plan and the residual skip index
    torch.cuda.synchronize()
    first_scan_s = time.time() - t0
    for _ in range(args.warmup):
        r, tot_arcs, tot_chk = step()
    resident = free0 - torch.cuda.mem_get_info(dev)[0]
    # ---- correctness gate (untimed) ----
    want_arcs = arcs_graph if scaling == "strong" else arcs_graph * world
    assert tot_arcs == want_arcs, (tot_arcs, want_arcs)
    if scaling == "weak":
        assert r["arcs"] == arcs_graph and r["nodes"] == n_graph, (r, arcs_graph, n_graph)
    if not args.no_verify:
        # Tiles of this rank's node range (the first, a middle and the last one, every base of the mosaic at least once, the tiles on
        # either side of nodes 2^31 and 2^32), scanned through the very handle that is timed, must give the checksum the CPU oracle
        # computes for those nodes; a graph from disk is gated on sample node ranges the same way.
        from oracle import bvg_oracle as O
        nb = rank * n_graph if scaling == "weak" else 0
        if sts:
            n0 = args.base_nodes; K = len(sts)
            ogs = [O.Graph.from_memory(O.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets) for st in sts]
            j_lo, j_hi = (lo + n0 - 1) // n0, hi // n0                   # whole tiles inside [lo, hi)
            jb, jb2 = (1 << 31) // n0, (1 << 32) // n0
            picks = {j_lo, (j_lo + j_hi) // 2, j_hi - 1} | ({jb - 1, jb, jb2 - 1, jb2} & set(range(j_lo, j_hi)))
            picks |= {j for j in range(j_lo, min(j_hi, j_lo + K))} | {j for j in range(max(j_lo, j_hi - K), j_hi)}
            for j in sorted(picks) if j_hi > j_lo else []:
                ro = ogs[j % K].scan(0, n0, node_base=nb + j * n0, threads=threads)
                rg = g.scan(j * n0, (j + 1) * n0)
                assert (rg["arcs"], rg["chk"]) == (ro["arcs"], ro["chk"]), "GPU scan of tile %d disagrees with the CPU oracle" % j
            # ... and ONE gate that does not rest on the checksum (which is linear in the successors: two errors inside one node that cancel in their sum are invisible to it):
            # the middle tile MATERIALISED through bvg_decode_range, every successor compared with the oracle's (BV coding is translation invariant: tile j = its base + j * n0)
            if j_hi > j_lo and world == 1:                                # (one rank: the N > 1 path cannot be rehearsed on hardware here, and its gate stays what the rehearsals ran)
                j = (j_lo + j_hi) // 2
                gv = g.copy(); gv.set_tuning(block_bits=args.block_bits); gv.set_node_base(g.node_base())   # (a flyweight: what the timed handle has learned about its blocks -- per mode, scan or materialise -- stays untouched)
                deg, succ = gv.decode_range(j * n0, (j + 1) * n0)
                del gv
                odeg, osucc = ogs[j % K].decode_range(0, n0)
                assert np.array_equal(deg, odeg) and np.array_equal(succ, osucc + (nb + j * n0)), "materialised successors of tile %d disagree with the CPU oracle" % j
                del deg, succ, odeg, osucc
            del ogs
        else:
            og = O.Graph.load(args.basename)
            span = max(1, min(hi - lo, 1 << 18))
            for a0 in sorted({lo, lo + (hi - lo - span) // 2, hi - span}):
                ro = og.scan(a0, a0 + span, node_base=nb, threads=threads)
                rg = g.scan(a0, a0 + span)
                assert (rg["arcs"], rg["chk"]) == (ro["arcs"], ro["chk"]), "GPU scan of nodes [%d, %d) disagrees with the CPU oracle" % (a0, a0 + span)
            del og
    if scaling == "strong" and world > 1 and args.verify_whole and rank == 0:
        # optional (it makes rank 0 index and scan the whole replica, which the timed configuration never does): the reduced pair against the ONE-PIECE scan
        rw = g.scan(0, n_graph)
        assert (rw["arcs"], rw["chk"]) == (tot_arcs, tot_chk), "the shards' reduced {arcs, chk} differ from the one-piece scan"
    if scaling == "strong" and world > 1 and not args.no_verify:
        # Every rank gates ITS OWN shard (no rank scans -- or indexes -- the whole replica): the tiles above against the CPU oracle, and
        # the shard as a whole against the sum of its pieces scanned one by one through the same handle (additivity: the checksum is a
        # plain sum over arcs, so the pieces of [lo, hi) must add up to the shard's own {arcs, chk}); the shards' sum is the all-reduce.
        cuts = sorted({lo, hi} | {lo + (hi - lo) * i // 7 for i in range(1, 7)})
        pa = pc = 0
        for a0, a1 in zip(cuts[:-1], cuts[1:]):
            rp = g.scan(a0, a1); pa += rp["arcs"]; pc = (pc + rp["chk"]) & 0xFFFFFFFFFFFFFFFF
        assert (pa, pc) == (int(r["arcs"]), int(r["chk"])), "rank %d: the pieces of the shard do not add up to the shard" % rank

    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        r, tot_arcs, tot_chk = step()
        kernel_ms.append(r["kernel_ms"])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    my_k_ms = float(np.mean(kernel_ms))
    if dist is not None:
        elapsed = S.allreduce_max(elapsed, device=cuda)
        k_ms = S.allreduce_max(my_k_ms, device=cuda)
        idx_all, lean_all = S.allreduce_scan(int(r["index_entries"]), int(r["lean_blocks"]), device=cuda)   # what every rank scanned with (untimed bookkeeping)
        per_rank_ms = S.allgather_float(my_k_ms, rank, world, device=cuda)                                  # shard imbalance: the slowest rank sets the step
        per_rank_idx = S.allgather_float(float(r["index_entries"]), rank, world, device=cuda)
    else:
        k_ms = my_k_ms
        idx_all, lean_all = int(r["index_entries"]), int(r["lean_blocks"])
        per_rank_ms, per_rank_idx = [my_k_ms], [float(r["index_entries"])]

    # ---- the index-less steady state (behind the timed region): the same scan through a flyweight that neither builds nor reads the
    # residual skip index (bvg_tuning.no_index) -- what a consumer that scans a graph ONCE gets -- and how many scans the index takes to pay back
    no_index = None
    if not args.no_index_leg:
        g0 = g.copy(); g0.set_tuning(block_bits=args.block_bits, no_index=True); g0.set_node_base(g.node_base())
        rn = g0.scan(lo, hi)                                              # (tier learning of the index-less handle)
        assert (rn["arcs"], rn["chk"]) == (int(r["arcs"]), int(r["chk"])), "the index-less scan disagrees with the indexed one"
        torch.cuda.synchronize(); tn0 = time.perf_counter()
        for _ in range(3):
            rn = g0.scan(lo, hi)
        torch.cuda.synchronize(); tn = (time.perf_counter() - tn0) / 3
        no_index = {"s_per_step": tn, "index_entries": int(rn["index_entries"]), "lean_blocks": int(rn["lean_blocks"])}
        del g0
        if dist is not None:
            no_index["s_per_step"] = S.allreduce_max(tn, device=cuda)

    # ---- "marks only" (behind the timed region): what a consumer short of HBM gets -- a SECOND handle on the same synthetic stream (its own plan and index: an index belongs to
    # its graph) with bvg_tuning.no_index = 2: the first scan validates and marks the blocks (the lean scan kernel takes them) but keeps entries only for lists of >= 4 096
    # residuals: ~0.03 % of the stream instead of ~50 %.  Mosaic workloads on one rank only (the second graph takes another stream's worth of HBM for the leg).
    marks = None
    if not args.no_index_leg and world == 1 and sts and (copies > 1 or len(bases) > 1):
        free1 = torch.cuda.mem_get_info(dev)[0]
        gm = W.mosaic(bases, copies)
        try:
            gm.set_tuning(block_bits=args.block_bits, no_index=2); gm.set_node_base(g.node_base())
            for _ in range(2):                                            # (validating pass = first scan; tier learning)
                rm = gm.scan(lo, hi)
            assert (rm["arcs"], rm["chk"]) == (int(r["arcs"]), int(r["chk"])), "the marks-only scan disagrees with the indexed one"
            torch.cuda.synchronize(); tm0 = time.perf_counter()
            for _ in range(3):
                rm = gm.scan(lo, hi)
            torch.cuda.synchronize(); tm = (time.perf_counter() - tm0) / 3
            marks = {"s_per_step": tm, "lean_blocks": int(rm["lean_blocks"]), "index_entries": int(rm["index_entries"]), "index_bytes": int(rm["index_bytes"]),
                     "resident": int(free1 - torch.cuda.mem_get_info(dev)[0])}
        finally:
            gm.close()

    # ---- the same scan at the reference's width (behind the timed region): BVGraph computes successors in `long`; the headline runs the 32-bit successor kernels, which is
    # lossless below 2^32 - 256 nodes (dtype "u32").  A second handle on the same synthetic stream with bvg_tuning.force_wide = 1 (its own plan and its own skip index with
    # 64-bit values: an index belongs to its width) scans through the 64-bit-id path -- the scan kernel on lists of ids relative to a per-block base, the checking kernels on
    # 64-bit lists: what a graph beyond 2^32 - 256 nodes runs.  Same checksum, its own rate.  Mosaic workloads on one rank only.
    wide = None
    if not args.no_wide_leg and n_graph <= 0xFFFFFF00 and world == 1 and sts and (copies > 1 or len(bases) > 1):
        gw = W.mosaic(bases, copies)
        try:
            gw.set_tuning(block_bits=args.block_bits, force_wide=True); gw.set_node_base(g.node_base())
            for _ in range(2):                                            # (index build of the wide form + tier learning)
                rw = gw.scan(lo, hi)
            assert (rw["arcs"], rw["chk"]) == (int(r["arcs"]), int(r["chk"])), "the 64-bit-id scan disagrees with the 32-bit one"
            torch.cuda.synchronize(); tw0 = time.perf_counter()
            for _ in range(3):
                rw = gw.scan(lo, hi)
            torch.cuda.synchronize(); tw = (time.perf_counter() - tw0) / 3
            wide = {"s_per_step": tw, "lean_blocks": int(rw["lean_blocks"]), "slow_blocks": int(rw["slow_blocks"]), "index_entries": int(rw["index_entries"])}
        finally:
            gw.close()

    # ---- a REAL web graph beside the stand-in (behind the timed region; never part of `value`): the reference's own fixture cnr-2000 (LAW; 9.9 arcs per node, 24 % empty
    # nodes, reference chains of depth 3), its stream repeated on the device to 4 GiB, gated tile by tile against the CPU oracle, 3 warm-up + 8 timed scans
    real = None
    if not args.no_real_leg and world == 1 and args.shape == "eu15" and not args.basename and os.path.exists(GOLDEN + ".graph"):
        try:
            real = real_graph_leg(W, torch, dev, threads)
        except Exception as ex:                                             # (the second workload must never take the headline line with it)
            real = {"error": repr(ex)}

    if rank == 0:
        edges_per_s = tot_arcs * args.steps / elapsed
        gbytes = r["graph_bytes"]                                        # this rank's shard (strong) or graph (weak)
        whole_bytes = (sum(int(st.offsets[-1]) for st in sts) * copies + 7) // 8 if sts else os.path.getsize(args.basename + ".graph")
        total_gbytes = whole_bytes * (1 if scaling == "strong" else world)
        gbs = gbytes / (k_ms * 1e-3) / 1e9                               # per-GPU rate of the dominant kernel family
        steady_s = elapsed / args.steps
        out = {
            "metric": "decoded edges/s, full sequential successor scan", "value": edges_per_s, "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": steady_s * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "u32" if n_graph <= 0xFFFFFF00 else "u64",
            "data": ("real: %s" % args.basename) if args.basename else ("real LAW graph cnr-2000 (the reference's fixture), its stream repeated on the device" if kind == "golden" else "synthetic"),
            "config": {"workload": wl + (" [stand-in: no LAW dataset on the box]" if args.shape.startswith("eu15") and not args.basename else ""), "shape": args.basename or args.shape,
                       "nodes": n_graph * (world if scaling == "weak" else 1), "arcs": tot_arcs, "graph_bytes": total_gbytes,
                       "nodes_per_gpu": hi - lo, "arcs_per_gpu": int(r["arcs"]), "graph_bytes_per_gpu": gbytes,
                       "bits_per_link": 8.0 * gbytes / max(int(r["arcs"]), 1), "tiles": copies * max(len(sts), 1), "distinct_tiles": max(len(sts), 1), "base_nodes": args.base_nodes,
                       "structure": structure_of(sts),
                       "sharding": ("%d arc-balanced node-range shard(s) of one graph, a replica per GPU" % world if scaling == "strong" else "%d graph(s), one per GPU, ids shifted" % world)
                                   + "; RCCL all-reduce of {arcs,chk} only",
                       "collective": {"backend": (args.backend + (" (RCCL)" if args.backend == "nccl" else "")) if dist is not None else None,
                                      "world_size": (dist.get_world_size() if dist is not None else 1), "one_device": bool(args.one_device)}},
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "traffic": None, "traffic_source": None,
                         "kernel": ("bvg::scan_kernel (tier 0 and, with larger pools, the big-LDS classes: %d of %d blocks of the last scan)" % (r["lean_blocks"], r["lean_blocks"] + r["slow_blocks"]) if r.get("lean_blocks") else "bvg::rows_kernel<%s,scan,tasks> (tier 0 and, with larger pools, the big-LDS classes)" % ("u32" if n_graph <= 0xFFFFFF00 else "u64"))
                                   + ", rows_kernel / giant_kernel / decode_kernel<slow> for the other blocks and reduce_acc_kernel launched beside it: hipEvent time of one scan on the handle's stream", "kernel_ms": k_ms,
                         "algorithmic_bytes_per_launch": gbytes, "index_bytes_per_launch": r["index_bytes"],
                         # the bytes the kernels really have to read: the stream AND the index beside it (packed offsets, block plan, skip entries)
                         "frac_with_index": (gbytes + r["index_bytes"]) / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "checksum": "%016x" % tot_chk, "arcs": tot_arcs, "slow_blocks": r["slow_blocks"],
            "index": {"skip_entries_rank0": int(r["index_entries"]), "skip_entries_all_ranks": idx_all, "lean_blocks_rank0": int(r["lean_blocks"]), "lean_blocks_all_ranks": lean_all},
            "index_build_s": max(first_scan_s - steady_s, 0.0), "hbm_resident_bytes": int(resident),
            "per_rank_kernel_ms": {"min": min(per_rank_ms), "mean": float(np.mean(per_rank_ms)), "max": max(per_rank_ms), "all": per_rank_ms},
            "imbalance": max(per_rank_ms) / max(float(np.mean(per_rank_ms)), 1e-9),
            "per_rank_index_entries": [int(v) for v in per_rank_idx],
            "host": {"generate_s": gen_s, "upload_s": upload_s, "tile_s": tile_s, "first_scan_s": first_scan_s},
        }
        if real is not None:
            out["real_graph"] = real
        if no_index is not None:
            # steady rate WITHOUT the index, and the number of indexed scans after which building it has paid for itself
            v0 = tot_arcs / no_index["s_per_step"]
            out["value_no_index"] = v0
            out["no_index"] = {"ms_per_step": no_index["s_per_step"] * 1e3, "steps": 3, "index_entries_used": no_index["index_entries"], "lean_blocks": no_index["lean_blocks"],
                               "how": "bvg_copy() flyweight with bvg_tuning.no_index = 1: same stream, plan and offsets in HBM, no skip entries, no validation marks (every block on the checking kernels)"}
            gain = no_index["s_per_step"] - steady_s
            out["index_break_even_scans"] = (out["index_build_s"] / gain) if gain > 0 else None
        if marks is not None:
            out["value_marks_only"] = tot_arcs / marks["s_per_step"]
            out["marks_only"] = {"ms_per_step": marks["s_per_step"] * 1e3, "steps": 3, "lean_blocks": marks["lean_blocks"], "index_entries": marks["index_entries"],
                                 "index_bytes_per_launch": marks["index_bytes"], "hbm_resident_bytes": marks["resident"],
                                 "how": "a second handle on the same stream with bvg_tuning.no_index = 2: validation marks (one byte per block) and skip entries for lists of >= 4 096 residuals only; "
                                        "the lean scan kernel with one lane per residual list"}
        if wide is not None:
            out["value_wide"] = tot_arcs / wide["s_per_step"]
            out["wide"] = {"ms_per_step": wide["s_per_step"] * 1e3, "steps": 3, "dtype": "u64 ids (lists relative to a per-block base in the scan kernel; 64-bit lists on the checking kernels)",
                           "lean_blocks": wide["lean_blocks"], "slow_blocks": wide["slow_blocks"], "index_entries": wide["index_entries"],
                           "how": "a second handle on the same stream with bvg_tuning.force_wide = 1 (its own index with 64-bit values): the path a graph beyond 2^32 - 256 nodes takes; same {arcs, chk} asserted"}
        t = measured_pmc(args.basename or args.shape, copies, args.base_nodes, world, scaling)
        if t:
            out["roofline"]["traffic"], out["roofline"]["traffic_source"] = t["hbm_bytes_per_launch"], t.get("source")
            if t.get("valu_per_arc"):
                # the fraction that matters for THIS kernel: vector-instruction issue.  VALU wave-instructions per arc from the
                # PMC pass on file for this very workload and kernel revision x the arcs/s measured NOW, against the issue peak measured on the card.
                ach = t["valu_per_arc"] * edges_per_s / max(world, 1)
                # (round 6: `frac` is measured against the FULL-rate issue peak -- one wave-instruction per 1.07 ns and SIMD, what v_add / and / lshr reach on the card; rounds 3-5
                #  divided by the half-rate peak, which flattered: that figure stays as frac_half_rate_peak)
                out["roofline_valu"] = {"bound": "valu-issue", "achieved": ach, "peak": VALU_PEAK_FULL_RATE, "unit": "wave-instr/s", "frac": ach / VALU_PEAK_FULL_RATE,
                                        "frac_full_rate": ach / VALU_PEAK_FULL_RATE, "frac_half_rate_peak": ach / VALU_PEAK_WINSTR_PER_S, "peak_half_rate_ops": VALU_PEAK_WINSTR_PER_S,
                                        "valu_per_arc": t["valu_per_arc"], "salu_per_arc": t.get("salu_per_arc"), "lds_per_arc": t.get("lds_per_arc"), "active_lanes": t.get("active_lanes"),
                                        "floor_valu_per_arc": VALU_FLOOR_PER_ARC, "x_floor": t["valu_per_arc"] / VALU_FLOOR_PER_ARC,
                                        "edges_per_s_at_floor": VALU_PEAK_WINSTR_PER_S / VALU_FLOOR_PER_ARC, "source": t.get("valu_source"),
                                        "peak_full_rate_ops": VALU_PEAK_FULL_RATE,
                                        "note": "peak = 1 024 SIMDs / 1.07 ns, the issue time of v_add / sub / and / or / xor / lshr / mov measured on the card (profiles/r04_valu_rates2.txt); shifts left, "
                                                "three-operand integer forms, multiplies, compares and selects take 1.83 ns (frac_half_rate_peak divides by that peak). Since round 6 the kernel is no longer "
                                                "purely issue-bound: with fewer instructions per arc a wavefront's own dependent chains (LDS round trips of the bit cursors, cross-lane dealing) show"}
        if not args.no_cpu_baseline and world == 1:                      # (rank 0 at N = 1 only: at N > 1 the other ranks would wait for it)
            out["cpu_baseline"] = cpu_baseline(sts, bases, args.basename, effective_cpus(threads), args.cpu_gib)   # one thread per CPU the box really grants
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def self_launch(n):
    """Runs this very command line under `python -m torch.distributed.run --nproc-per-node n` in a child process and returns its
    exit code.  Rank 0 of the child prints the JSON line to the inherited stdout."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def measured_pmc(shape, tiles, base_nodes, world, scaling):
    """Counter figures per launch from a PMC pass of THIS workload on THIS kernel revision, if one is on file (profiles/traffic.json:
    HBM bytes, VALU / SALU / LDS wave-instructions per arc, active lanes -- with the command they came from); never a figure from
    another configuration or from an earlier round's kernels."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:
        return None
    src = kernel_source_id()
    for e in rec.get("runs", []):
        if e.get("rev") != KERNEL_REV or e.get("src_id") != src:
            continue
        if (e.get("shape"), e.get("tiles"), e.get("base_nodes"), e.get("n_gpus", 1), e.get("scaling", "weak")) == (shape, tiles, base_nodes, world, scaling if world > 1 else e.get("scaling", "weak")):
            return e
    return None


class _Store:
    """What bench.py needs of a stored graph (tooling.Stored's fields)."""

    def __init__(self, params, graph, offsets, stats):
        self.params, self.graph, self.offsets, self.stats = params, graph, offsets, stats


def golden_store(basename):
    """The reference's fixture as a stored graph: properties parsed by the library's own parser, the .graph bytes verbatim, the
    offsets decoded from the .offsets file (bvg_decode_offsets, host side)."""
    import numpy as np
    import webgraph_big_amd as W
    p = W.parse_properties(open(basename + ".properties").read())
    graph = np.fromfile(basename + ".graph", dtype=np.uint8)
    offs = W.decode_offsets(np.fromfile(basename + ".offsets", dtype=np.uint8), int(p.nodes), int(p.offset_coding))
    return _Store(p, graph, offs, {"arcs": int(p.arcs)})


def save_stores(prefix, sts):
    import numpy as np
    for i, st in enumerate(sts):
        np.save("%s_%d_g.npy" % (prefix, i), st.graph); np.save("%s_%d_o.npy" % (prefix, i), st.offsets)
        with open("%s_%d_p.bin" % (prefix, i), "wb") as f:
            f.write(bytes(st.params))
        with open("%s_%d_s.json" % (prefix, i), "w") as f:
            json.dump({k: int(v) for k, v in st.stats.items()}, f)
    with open(prefix + "_n", "w") as f:
        f.write(str(len(sts)))


def load_stores(prefix):
    import numpy as np
    import webgraph_big_amd as W
    out = []
    for i in range(int(open(prefix + "_n").read())):
        p = W.Params.from_buffer_copy(open("%s_%d_p.bin" % (prefix, i), "rb").read())
        out.append(_Store(p, np.load("%s_%d_g.npy" % (prefix, i)), np.load("%s_%d_o.npy" % (prefix, i)), json.load(open("%s_%d_s.json" % (prefix, i)))))
    return out


def remove_stores(prefix):
    import glob
    for f in glob.glob(prefix + "_*"):
        try:
            os.remove(f)
        except OSError:
            pass


def effective_cpus(threads):
    """CPUs this process may really use: the affinity mask and the cgroup CPU quota bound the thread count."""
    n = threads
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(round(int(txt[0]) / int(txt[1])))))
            else:
                q = int(txt[0]); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(round(q / per))))
            break
        except Exception:
            continue
    return n


def real_graph_leg(W, torch, dev, threads, gib=4.0, steps=8):
    """`--shape cnr` in small, inside the default run: cnr-2000 from tests/golden/ tiled on the device, first scan (index build), 3 warm-up scans, `steps` timed
    scans (hipEvent time of the kernels and wall clock), three tiles gated against the CPU oracle."""
    from oracle import bvg_oracle as O
    st = golden_store(GOLDEN)
    n0 = int(st.params.nodes)
    copies = max(1, min(int(gib * (1 << 30) / max(len(st.graph), 1)), ((1 << 31) - 1) // n0))
    base = W.BVGraph.from_memory(st.params, st.graph, st.offsets, device=dev)
    g = None
    try:                                                                  # (the handles are closed whatever fails: a second 4 GiB graph must not outlive its leg)
        g = W.mosaic([base], copies)
        return _real_graph_scans(W, torch, g, st, n0, copies, threads, steps, O)
    finally:
        if g is not None:
            g.close()
        base.close()


def _real_graph_scans(W, torch, g, st, n0, copies, threads, steps, O):
    n = g.num_nodes()
    r = g.scan()
    for _ in range(3):
        r = g.scan()
    torch.cuda.synchronize(); t0 = time.perf_counter(); kms = 0.0
    for _ in range(steps):
        r = g.scan(); kms += r["kernel_ms"]
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    og = O.Graph.from_memory(O.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)
    for j in sorted({0, copies // 2, copies - 1}):
        ro = og.scan(0, n0, node_base=j * n0, threads=threads); rg = g.scan(j * n0, (j + 1) * n0)
        assert (rg["arcs"], rg["chk"]) == (ro["arcs"], ro["chk"]), "GPU scan of cnr-2000 tile %d disagrees with the CPU oracle" % j
    out = {"workload": "cnr-2000 (LAW; the reference's fixture tests/golden/cnr-2000.*, W=7 maxRef=3 minInterval=3 zeta3), its stream repeated %d times on the device" % copies,
           "value": r["arcs"] / dt, "unit": "edges/s", "ms_per_step": dt * 1e3, "kernel_ms": kms / steps, "steps": steps, "nodes": n, "arcs": int(r["arcs"]),
           "graph_bytes": int(r["graph_bytes"]), "roofline_frac": r["graph_bytes"] / (kms / steps * 1e-3) / 1e9 / HBM_PEAK_GBS, "lean_blocks": int(r["lean_blocks"]),
           "gated": "tiles 0, %d, %d against the CPU oracle" % (copies // 2, copies - 1), "note": "same command as `python bench.py --shape cnr --target-gib 4`; not part of `value`"}
    return out


def structure_of(sts):
    """How the workload's lists are built, in the reference's own terms (avgref / avgdist as BVGraph.store writes them, BVG:2280-2281,2500-2501; arc provenance as in SURVEY
    Appendix B), next to the same figures of the one real LAW graph in the tree -- so that a reader can see how far the stand-in is from a crawl."""
    cnr = {"avgref": 1.38, "avgdist": 1.74, "copied": 0.663, "intervals": 0.113, "residuals": 0.225, "arcs_per_node": 9.88, "bits_per_link": 3.56,
           "source": "tests/golden/cnr-2000.properties (the reference's fixture) and SURVEY.md Appendix B"}
    tot = {k: 0 for k in ("arcs", "copied", "intervalised", "residual", "tot_ref", "tot_dist")}
    nodes = 0
    for st in sts or []:
        s = getattr(st, "stats", None)
        if not s or "copied" not in s:                                    # (a graph from disk: only what its .properties say)
            return {"cnr-2000": cnr}
        for k in tot:
            tot[k] += int(s.get(k, 0))
        nodes += int(st.params.nodes)
    if not nodes or not tot["arcs"]:
        return {"cnr-2000": cnr}
    a = float(tot["arcs"])
    return {"avgref": tot["tot_ref"] / nodes, "avgdist": tot["tot_dist"] / nodes, "copied": tot["copied"] / a, "intervals": tot["intervalised"] / a, "residuals": tot["residual"] / a,
            "arcs_per_node": a / nodes, "cnr-2000": cnr}


def reference_java(basename):
    """The reference itself, timed beside the GPU when the box can run it: `java -cp $BVG_REFERENCE_CP it.unimi.dsi.big.webgraph.test.SpeedTest
    <basename>` (test/SpeedTest.java:117-146: single-threaded sequential scan, 3 warm-up + 10 timed) needs a JVM, the reference's classes with their dsiutils / fastutil /
    sux4j jars on BVG_REFERENCE_CP, and a graph on disk.  None of them exists in this image (no java, no jars, no network): the probe says so instead of staying silent."""
    import shutil
    import subprocess
    java, cp = shutil.which("java"), os.environ.get("BVG_REFERENCE_CP")
    if not java:
        return "not available: no `java` on PATH (run-time probe); the reported baseline is the C port of the same decode path"
    if not cp or not basename:
        return "not run: a JVM exists (%s) but %s" % (java, "BVG_REFERENCE_CP does not name the reference's classes and jars" if not cp else "the workload is synthetic (SpeedTest takes a basename on disk)")
    try:
        r = subprocess.run([java, "-server", "-cp", cp, "it.unimi.dsi.big.webgraph.test.SpeedTest", basename], capture_output=True, text=True, timeout=1800)
        return {"cmd": "java -server -cp $BVG_REFERENCE_CP it.unimi.dsi.big.webgraph.test.SpeedTest " + basename, "rc": r.returncode, "stdout_tail": r.stdout[-1500:], "stderr_tail": r.stderr[-500:]}
    except Exception as ex:                                               # a broken class path must not take the bench line with it
        return "failed to run: %r" % (ex,)


def cpu_baseline(sts, bases_gpu, basename, threads, gib):
    """The CPU oracle -- an unoptimised C restatement (PORT) of the reference's decode path, per-node mallocs and all; no JVM exists
    here, so not the reference itself and no statement about the Java's speed -- timed on this box's host cores over a `gib`-GiB prefix
    of the workload (whole cycles of the mosaic, concatenated on the host exactly as bvg_mosaic does on the device; a graph from disk:
    a prefix of its nodes), node ranges split as ImmutableGraph.splitNodeIterators does."""
    import numpy as np
    from oracle import bvg_oracle as O
    import tooling as T
    if sts:
        k = max(1, int(round(gib * (1 << 30) / max(sum(len(st.graph) for st in sts), 1))))
        ts = T.mosaic_host(sts, k)
        og = O.Graph.from_memory(O.Params(**ts.params.as_dict()), ts.graph.tobytes(), ts.offsets)
        n = ts.params.nodes
        what = "first %d cycle(s) of the workload (%d tiles, " % (k, k * len(sts))
        gbytes = len(ts.graph)
    else:
        og = O.Graph.load(basename)
        n_all = og.num_nodes()
        gsize = os.path.getsize(basename + ".graph")
        n = max(1, min(n_all, int(n_all * min(1.0, gib * (1 << 30) / max(gsize, 1)))))
        what = "first %d of %d nodes of %s (" % (n, n_all, os.path.basename(basename))
        gbytes = int(gsize * n / max(n_all, 1))
    reps, tm, first, r1 = 0, 0.0, None, None
    while (tm < 10.0 and reps < 64) or reps < 2:                          # sustained rate: a CPU quota lets the first burst run faster
        t0 = time.perf_counter(); r1 = og.scan(0, n, threads=threads); dt = time.perf_counter() - t0
        first = dt if first is None else first
        tm += dt; reps += 1
    tm /= reps
    gate = False
    if sts:                                                              # also gate the GPU result on it: the first tile of the workload must produce the same checksum
        n0 = sts[0].params.nodes
        r0 = og.scan(0, n0, threads=threads)
        bases_gpu[0].set_node_base(0)
        rg = bases_gpu[0].scan()
        assert (rg["arcs"], rg["chk"]) == (r0["arcs"], r0["chk"]), "GPU scan disagrees with the CPU oracle"
        gate = True
    # single-thread figure on a bounded sample (~5 s)
    sample = max(1, min(n, int(n * min(1.0, 5.0 / max(tm * threads, 1e-3)))))
    t0 = time.perf_counter(); r2 = og.scan(0, sample, threads=1); t1 = time.perf_counter() - t0
    return {"value": r1["arcs"] / tm, "unit": "edges/s", "cores": threads, "threads": threads, "value_first_scan": r1["arcs"] / first, "kind": "port",
            "reference_java": reference_java(basename),
            "sample": what + "%d nodes, %d arcs, %.2f GiB of .graph), mean of %d scans with %d threads over contiguous node ranges; 1 thread on the first %d nodes: %.3g edges/s; the port is an unoptimised restatement (a malloc per node), not the reference's Java"
                      % (n, r1["arcs"], gbytes / (1 << 30), reps, threads, sample, r2["arcs"] / max(t1, 1e-9)),
            "value_1thread": r2["arcs"] / max(t1, 1e-9), "gpu_matches_oracle": gate}


if __name__ == "__main__":
    main()
