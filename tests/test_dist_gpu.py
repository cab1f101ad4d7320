"""-m gpu: the data-parallel reducer on the HIP path over RCCL (backend "nccl").

* one rank (every GPU box): a 1-rank RCCL group with ESVIT_FORCE_REDUCER=1 runs the bucket slots, the post-accumulate hooks and
  the asynchronous all_reduce(AVG) for real; three trainer steps leave student and teacher equal to the same steps without a
  reducer -- ragged route, per-group schedule (two gradient contributions per parameter) and CvT;
* two ranks (skipped on 1-GPU boxes): averaged gradients = mean of the per-rank gradients.
The workers are separate processes (tests/dist_gpu_worker.py): a process group must not leak into the pytest process."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_gpu_worker.py")


def _run(mode, world, port, tmp_path):
    path = str(tmp_path / "res.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, WORKER, mode, str(r), str(world), str(port), path], env=env, cwd=ROOT,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [json.load(open(path + ".%d" % r)) for r in range(world)]


def test_reducer_one_rank_rccl_equals_plain_step(lib_built, tmp_path):
    (res,) = _run("one_rank", 1, 29731, tmp_path)
    print("REDUCER one-rank RCCL:", json.dumps(res))
    bf = res.pop("ragged_bf16_payload")
    # AdamW's normalised step moves an element by at most lr = 5e-4 per step, so two runs whose (near-zero) gradient elements round to
    # opposite signs end at most 2 * 3 * lr apart after three steps (observed 1.2e-3 .. 1.5e-3); the bf16 nano step itself carries 3e-3
    # of loss noise
    assert bf["reducer_on"] and bf["max_abs_param_diff"] < 2 * 3 * 5e-4 and bf["loss_diff"] < 2e-2, bf
    for kind, r in res.items():
        assert r["reducer_on"] and r["reducer_off"] and r["buckets"] >= 1, (kind, r)
        assert r["losses_equal"], (kind, r)
        # (bit-identical unless the per-tensor clip is active: its squared norms are summed with float atomics)
        assert r["bit_identical"] or r["max_rel"] < 2e-6, (kind, r)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_reducer_two_ranks_rccl(lib_built, tmp_path):
    res = _run("world2", 2, 29732, tmp_path)
    for rank, r in enumerate(res):
        for kind, v in r.items():
            assert v["averaged"] and v["buckets"] >= 2, (rank, kind, v)
