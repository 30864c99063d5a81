"""Multi-GPU sharding of the scan: one process per GPU, node ranges as shards, and the path's only
collective — a sum all-reduce of {arcs, checksum} (RCCL over xGMI when the backend is "nccl").

Mirrors ImmutableGraph.splitNodeIterators (ImmutableGraph.java:405-436): contiguous node ranges, no
data exchanged between shards (each shard re-derives its own halo from its copy/slice of .graph)."""
import numpy as np


def split_nodes(n, k):
    """ceil(n/k)-sized contiguous ranges, exactly as ImmutableGraph.java:415-433; returns k (lo, hi) pairs."""
    m = -(-n // k) if n else 0
    out = []
    for i in range(k):
        lo = min(i * m, n)
        out.append((lo, min(lo + m, n)))
    return out


def u64_to_i64(v):
    return int(np.uint64(v & 0xFFFFFFFFFFFFFFFF).astype(np.int64))


def i64_to_u64(v):
    return int(np.int64(v).astype(np.uint64))


def allreduce_scan(arcs, chk, device=None, group=None):
    """Sum of per-shard {arcs, chk} mod 2^64 over all ranks (int64 two's-complement wrap == uint64 sum)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([u64_to_i64(arcs), u64_to_i64(chk)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    t = t.cpu()
    return i64_to_u64(t[0].item()), i64_to_u64(t[1].item())


def allreduce_max(value, device=None, group=None):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def allgather_float(value, rank, world, device=None, group=None):
    """Every rank's value, in rank order (a sum all-reduce of a vector with one slot per rank: works on every backend)."""
    import torch
    import torch.distributed as dist
    t = torch.zeros(world, dtype=torch.float64, device=device)
    t[rank] = float(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return [float(v) for v in t.cpu().tolist()]


def bounds_by_arcs(outdeg, k):
    """The rule of bvg_split_by_arcs on a host array of outdegrees: bounds[j] = first node whose cumulative outdegree (exclusive
    prefix) reaches j * ceil(arcs / k) (the skipTo() targets of algo/HyperBall.java:748-768); bounds[0] = 0, bounds[k] = n."""
    n = len(outdeg)
    cum = np.concatenate([[0], np.cumsum(np.asarray(outdeg, dtype=np.int64))]).astype(np.uint64)
    arcs = int(cum[-1])
    per = max(1, -(-arcs // k))
    b = np.searchsorted(cum[:n + 1], np.arange(k + 1, dtype=np.uint64) * np.uint64(per), side="left").astype(np.int64)
    b = np.minimum(b, n)
    b[0] = 0; b[k] = n
    return b


def sharded_scan(scan_range, bounds, rank, device=None, group=None, reduce=True):
    """One rank of the strong-scaling scan of ONE graph: this rank scans nodes [bounds[rank], bounds[rank+1]) with
    scan_range(lo, hi) -> {'arcs', 'chk', ...} (the HIP handle's scan on a GPU box, the oracle in the CPU tests) and the
    per-shard {arcs, chk} are summed over all ranks -- the path's only collective.  Returns (own result, arcs, chk)."""
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    r = scan_range(lo, hi)
    if not reduce:
        return r, int(r["arcs"]), int(r["chk"])
    arcs, chk = allreduce_scan(r["arcs"], r["chk"], device=device, group=group)
    return r, arcs, chk
