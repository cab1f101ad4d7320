"""Worker of tests/test_dist_gpu.py (one process per GPU, backend nccl = RCCL).  Run as
    python tests/dist_gpu_worker.py <mode> <rank> <world> <port> <result file>

modes
  one_rank   1-rank RCCL group with ESVIT_FORCE_REDUCER=1: the hook / bucket / all_reduce(AVG) machinery of
             engine.GradBucketReducer runs for real on the HIP path; three trainer steps must leave student and teacher parameters
             BIT-identical to the same steps without a reducer (averaging over one rank is the identity), for the ragged route, the
             per-group schedule (two contributions per parameter) and CvT.
  world2     two ranks on two GPUs: the averaged gradients of one step equal the mean of the per-rank gradients computed without a
             reducer, and both ranks hold identical parameters after the fused update.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def _models(kind, dev):
    import esvit_amd  # noqa: F401
    from tests import golden_utils as GU
    if kind == "cvt":
        from tests.test_composition_cpu import nano_cvt_pair
        student, teacher = nano_cvt_pair(dev)
        crops = lambda seed: [c.to(dev) for c in GU.make_crops(2, n_local=3, sizes=GU.NANO_CVT["sizes"], seed=seed)]  # noqa: E731
        ncrops = 5
    else:
        from tests.test_composition_cpu import nano_pair
        student, teacher = nano_pair()
        student, teacher = student.to(dev), teacher.to(dev)
        student.ragged_multi_crop = kind == "ragged"
        crops = lambda seed: [c.to(dev) for c in GU.make_crops(2, seed=seed)]  # noqa: E731
        ncrops = 10
    return student, teacher, crops, ncrops


def _steps(kind, dev, force, n_steps=3, payload="fp32"):
    """n trainer steps -> (student state, teacher state, losses, reducer enabled)"""
    import esvit_amd
    import esvit_amd.loss as L
    from esvit_amd import params as P
    from esvit_amd.engine import EsvitTrainer
    from tests import golden_utils as GU
    esvit_amd.set_precision("bf16")
    P.clear()
    if force:
        os.environ["ESVIT_FORCE_REDUCER"] = "1"
    else:
        os.environ.pop("ESVIT_FORCE_REDUCER", None)
    torch.manual_seed(0)
    student, teacher, crops, ncrops = _models(kind, dev)
    loss_fn = L.DDINOLoss(GU.NANO_HEAD["out_dim"], ncrops, 0.04, 0.07, 5, 10).to(dev)
    tr = EsvitTrainer(student, teacher, loss_fn, clip_grad=3.0, freeze_last_layer=0, teacher_stream=False, grad_payload=payload)
    losses = [tr.step(crops(900 + i), 5e-4, 0.04, 0.996, epoch=1).item() for i in range(n_steps)]
    torch.cuda.synchronize()
    out = ({k: v.detach().clone() for k, v in student.state_dict().items()}, {k: v.detach().clone() for k, v in teacher.state_dict().items()},
           losses, tr.reducer.enabled, len(tr.reducer.buckets))
    tr.reducer.close()
    return out


def one_rank(dev):
    res = {}
    for kind in ("ragged", "pergroup", "cvt"):
        s0, t0, l0, en0, _ = _steps(kind, dev, force=False)
        s1, t1, l1, en1, nb = _steps(kind, dev, force=True)
        same = all(torch.equal(s0[k], s1[k]) for k in s0) and all(torch.equal(t0[k], t1[k]) for k in t0)
        rel = max(((a[k].float() - b[k].float()).abs().max() / (a[k].float().abs().max() + 1e-12)).item()
                  for a, b in ((s0, s1), (t0, t1)) for k in a if a[k].numel())
        worst = max((((s0[k].float() - s1[k].float()).abs().max() / (s0[k].float().abs().max() + 1e-12)).item(), k) for k in s0 if s0[k].numel())
        res[kind] = dict(bit_identical=bool(same), max_rel=rel, losses_equal=l0 == l1, reducer_off=not en0, reducer_on=bool(en1), buckets=nb,
                         losses=[l0, l1], worst_param=worst[1])
    # bf16 gradient payload: the same three steps with the buckets rounded to bf16 on the wire -- parameters within bf16 rounding of
    # the gradients' effect (AdamW's normalised step is bounded by lr per element: compare against lr)
    s0, t0, l0, _, _ = _steps("ragged", dev, force=False)
    s2, t2, l2, en2, _ = _steps("ragged", dev, force=True, payload="bf16")
    dmax = max((s0[k].float() - s2[k].float()).abs().max().item() for k in s0 if s0[k].numel())
    res["ragged_bf16_payload"] = dict(reducer_on=bool(en2), max_abs_param_diff=dmax, loss_diff=max(abs(a - b) for a, b in zip(l0, l2)))
    return res


def world2(dev, rank, world):
    import esvit_amd
    import esvit_amd.loss as L
    from esvit_amd import params as P
    from esvit_amd.engine import GradBucketReducer
    from tests import golden_utils as GU
    esvit_amd.set_precision("fp32")
    res = {}
    for kind in ("ragged", "pergroup"):
        P.clear()
        student, teacher, _, ncrops = _models(kind, dev)
        K = GU.NANO_HEAD["out_dim"]

        def grads(r, reducer):
            crops = [c.to(dev) for c in GU.make_crops(1, seed=300 + r)]
            loss_fn = L.DDINOLoss(K, ncrops, 0.04, 0.04, 0, 1).to(dev)
            loss_fn._reduce_and_apply = lambda buf, apply: None
            for p in student.parameters():
                p.grad = None
            with torch.no_grad():
                t_out = teacher(crops[:2])
            loss = loss_fn(student(crops), t_out, 0, None)
            if reducer is not None:
                reducer.begin()
            loss.backward()
            if reducer is not None:
                reducer.finish()
            torch.cuda.synchronize()
            return {n: p.grad.detach().clone() for n, p in student.named_parameters() if p.grad is not None}

        red = GradBucketReducer(student, bucket_mb=0.25)
        got = grads(rank, red)
        red.close()
        acc = None
        for r in range(world):
            g = grads(r, None)
            acc = g if acc is None else {n: acc[n] + g[n] for n in g}
        ok = set(got) == set(acc) and all(torch.allclose(got[n], acc[n] / world, rtol=5e-4, atol=1e-6) for n in got)
        res[kind] = dict(averaged=bool(ok), buckets=len(red.buckets))
    return res


def main():
    mode, rank, world, port, path = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    res = one_rank(dev) if mode == "one_rank" else world2(dev, rank, world)
    with open(path + ".%d" % rank, "w") as fh:
        json.dump(res, fh)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
