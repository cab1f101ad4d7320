"""bench.py plumbing that can be checked without a GPU: `python bench.py --gpus N` re-executes itself as N ranks under
torch.distributed.run with a 127.0.0.1 rendezvous (the driver launches N > 1 that way itself; a user need not)."""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_respawns_under_torch_distributed_run(monkeypatch):
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    seen = {}

    def fake_execv(exe, argv):
        seen["exe"], seen["argv"] = exe, argv
        raise SystemExit(0)

    monkeypatch.setattr(os, "execv", fake_execv)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit):
        bench.main()
    a = seen["argv"]
    assert seen["exe"] == sys.executable and a[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in a and a[a.index("--nproc-per-node") + 1] == "4"
    assert a[a.index("--master-addr") + 1] == "127.0.0.1" and int(a[a.index("--master-port") + 1]) > 0
    i = a.index(os.path.join(ROOT, "bench.py"))
    assert a[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]  # the ranks see the caller's own flags


def test_bench_refuses_mismatched_world(monkeypatch):
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE" in str(e.value)
