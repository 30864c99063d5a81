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
