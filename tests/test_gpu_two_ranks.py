"""N > 1 in front of the driver's GPU run: `bench.py --gpus 2` on the box's one GPU -- bench.py's own launcher starts the two ranks as children of a
process that never touches the GPU, the ranks share cuda:0 and reduce {arcs, chk} over gloo -- must report the world size it ran with, the checksum of the one-rank
scan of the same workload, about half of the skip entries per rank and balanced shards (ImmutableGraph.java:405-436 is the split it mirrors, arc-balanced)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(*extra):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, BENCH, "--target-gib", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-index-leg", "--no-real-leg"] + list(extra),
                       capture_output=True, text=True, timeout=900, env=e)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]                                  # rank 0 prints ONE line
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_ranks_on_one_device_add_up_to_the_one_rank_scan():
    one = _run("--gpus", "1")
    two = _run("--gpus", "2", "--one-device", "--backend", "gloo")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert two["config"]["collective"] == {"backend": "gloo", "world_size": 2, "one_device": True}
    assert (two["checksum"], two["arcs"]) == (one["checksum"], one["arcs"])     # the shards' reduced pair IS the one-piece scan
    ent = two["per_rank_index_entries"]
    assert len(ent) == 2 and abs(ent[0] - ent[1]) <= 0.1 * sum(ent) and sum(ent) >= one["index"]["skip_entries_rank0"]   # each rank indexed its own half (+ the seam)
    assert two["imbalance"] < 1.3 and len(two["per_rank_kernel_ms"]["all"]) == 2
    assert two["config"]["arcs_per_gpu"] < 0.6 * two["arcs"]
