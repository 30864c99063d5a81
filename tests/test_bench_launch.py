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


def test_counter_figures_are_quoted_only_for_the_kernel_sources_they_were_measured_on(tmp_path, monkeypatch):
    """bench.py quotes `roofline.traffic` / `roofline_valu` from profiles/traffic.json only while webgraph-big_amd/csrc hashes to the sources the PMC pass ran on
    (a stale entry must not decorate a new kernel)."""
    import importlib
    import json
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    sid = bench.kernel_source_id()
    assert len(sid) == 16 and sid == bench.kernel_source_id()
    rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    mine = [e for e in rec["runs"] if e.get("rev") == bench.KERNEL_REV]
    assert mine and all("src_id" in e for e in mine)
    e = mine[0]
    got = bench.measured_pmc(e["shape"], e["tiles"], e["base_nodes"], 1, "weak")
    assert (got is not None) == (e["src_id"] == sid)                            # quoted iff measured on THESE sources
    monkeypatch.setattr(bench, "kernel_source_id", lambda: "0" * 16)
    assert bench.measured_pmc(e["shape"], e["tiles"], e["base_nodes"], 1, "weak") is None
