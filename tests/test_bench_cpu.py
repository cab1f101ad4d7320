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


def test_bench_line_schema_and_finite_loss_guard():
    """the ONE JSON line of bench.py (the driver's contract) carries every required field plus `losses_finite` / `refused_updates`,
    and a run whose timed steps produced a non-finite loss exits non-zero (rounds 2-4 reported NaN steps of the secondary
    configurations as measurements).  Checked on the source: the line is assembled on the GPU."""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    keys = set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Dict):
            ks = {k.value for k in node.keys if isinstance(k, ast.Constant) and isinstance(k.value, str)}
            if "metric" in ks and "value" in ks:
                keys = ks
    need = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "losses_finite", "refused_updates", "final_loss"}
    assert need <= keys, need - keys
    assert 'out["roofline"]' in src and 'out["cpu_baseline"]' in src
    # the guard: a non-finite loss makes the process exit with a non-zero code after printing the line
    guard = src[src.index("if not losses_finite:"):]
    assert "sys.exit(3)" in guard[:400]
