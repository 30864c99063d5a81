"""CPU: `python bench.py --gpus N` with no launcher in the environment starts N ranks itself (a child `torch.distributed.run`,
spawned before anything touches the GPU) and refuses to report a rank count it did not run with."""
import os
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _env():
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return e


def test_bench_starts_its_own_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--echo-ranks"], capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    import re
    assert sorted(re.findall(r"rank \d of 3", r.stdout)) == ["rank %d of 3" % i for i in range(3)]


def test_one_rank_needs_no_launcher():
    r = subprocess.run([sys.executable, BENCH, "--echo-ranks"], capture_output=True, text=True, timeout=120, env=_env())
    assert r.returncode == 0 and r.stdout.strip() == "rank 0 of 1"


def test_rank_count_mismatch_is_an_error():
    e = _env(); e.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--echo-ranks"], capture_output=True, text=True, timeout=120, env=e)
    assert r.returncode != 0 and "launcher started 1 rank" in r.stderr
