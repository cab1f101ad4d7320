"""world_size-2 gloo tests (CPU): the data-parallel pieces of the step -- bucketed, hook-driven gradient averaging
(engine.GradBucketReducer) and the single fused all-reduce of the two loss centres (DDINOLoss.update_center) -- give
the same result as one process seeing the whole batch."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _init(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // world))  # the ranks share the host cores
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _reducer_worker(rank, world, port, out):
    _init(rank, world, port)
    from esvit_amd.engine import GradBucketReducer
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(40, 300), torch.nn.GELU(), torch.nn.Linear(300, 300), torch.nn.Linear(300, 7))
    net[3].weight.requires_grad_(False)  # a frozen parameter must be skipped
    red = GradBucketReducer(net, bucket_mb=0.2)  # 0.2 MiB buckets -> several buckets
    assert red.enabled and len(red.buckets) >= 2
    g = torch.Generator().manual_seed(100)
    xs = torch.randn(world, 16, 40, generator=g)
    for step in range(2):  # two steps: grads are re-pointed at bucket views, must keep working
        net.zero_grad(set_to_none=True)
        red.begin()
        (net(xs[rank]) ** 2).mean().backward()
        red.finish()
    got = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    # reference: average of the per-rank gradients computed without the reducer
    ref = torch.nn.Sequential(torch.nn.Linear(40, 300), torch.nn.GELU(), torch.nn.Linear(300, 300), torch.nn.Linear(300, 7))
    ref.load_state_dict(net.state_dict())
    ref[3].weight.requires_grad_(False)
    acc = None
    for r in range(world):
        ref.zero_grad(set_to_none=True)
        (ref(xs[r]) ** 2).mean().backward()
        gr = {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}
        acc = gr if acc is None else {n: acc[n] + gr[n] for n in gr}
    ok = set(got) == set(acc) and all(torch.allclose(got[n], acc[n] / world, rtol=1e-5, atol=1e-7) for n in got)
    out[rank] = bool(ok)
    dist.destroy_process_group()


def _reducer_bf16_worker(rank, world, port, out):
    """payload="bf16": the buckets travel as bf16 (half the bytes per link) and come back as the fp32 mean of the rounded values"""
    _init(rank, world, port)
    from esvit_amd.engine import GradBucketReducer
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(40, 300), torch.nn.GELU(), torch.nn.Linear(300, 300), torch.nn.Linear(300, 7))
    red = GradBucketReducer(net, bucket_mb=0.2, payload="bf16")
    assert red.enabled and len(red.buckets) >= 2 and all(w.dtype == torch.bfloat16 for w in red.wire)
    g = torch.Generator().manual_seed(100)
    xs = torch.randn(world, 16, 40, generator=g)
    per_rank = []
    for r in range(world):
        net.zero_grad(set_to_none=True)
        (net(xs[r]) ** 2).mean().backward()
        per_rank.append({n: p.grad.clone() for n, p in net.named_parameters()})
    net.zero_grad(set_to_none=True)
    red.begin()
    (net(xs[rank]) ** 2).mean().backward()
    red.finish()
    ok = True
    for n, p in net.named_parameters():
        want = sum(pr[n].bfloat16().float() for pr in per_rank) / world  # mean of the bf16-rounded per-rank gradients
        ok = ok and p.grad.dtype == torch.float32 and torch.allclose(p.grad, want, rtol=1e-2, atol=1e-6)
        exact = sum(pr[n] for pr in per_rank) / world
        ok = ok and (p.grad - exact).norm() <= 8e-3 * exact.norm() + 1e-9
    out[rank] = bool(ok)
    dist.destroy_process_group()


def _nano_step_worker(rank, world, port, out, ragged=True):
    """the whole nano step on two ranks through the product's host code (kernels replaced by their torch restatement):
    weight gradients are written straight into the reducer's bucket slots (params.grad_out), the rest is packed by the
    post-accumulate hook; the averaged gradients equal the mean of the per-rank gradients computed without a reducer"""
    _init(rank, world, port)
    import esvit_amd.functional as Fn
    import esvit_amd.loss as L
    import esvit_amd.params as P
    from esvit_amd.engine import GradBucketReducer
    from oracle import ops_ref
    from tests import golden_utils as GU
    from tests.test_composition_cpu import nano_pair
    for mod in (Fn, L, P):
        mod.ops = ops_ref
    ops_ref.set_act_dtype(torch.float32)
    K = GU.NANO_HEAD["out_dim"]

    def grads(student, teacher, r, reducer):
        crops = GU.make_crops(1, seed=300 + r)
        loss_fn = L.DDINOLoss(K, 10, 0.04, 0.04, 0, 1)
        loss_fn._reduce_and_apply = lambda buf, apply: None  # centres are not under test here (and the ranks would desynchronise)
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
        return {n: p.grad.detach().clone() for n, p in student.named_parameters() if p.grad is not None}

    student, teacher = nano_pair()
    student.ragged_multi_crop = ragged
    # the per-group schedule (ragged False) contributes to every backbone parameter once per resolution group: the bucket slot is
    # handed to the first contribution only (params.grad_out), autograd sums, the hook packs what did not land in the slot -- the
    # reducer overlaps for every schedule
    red = GradBucketReducer(student, bucket_mb=0.25)
    assert red.enabled and len(red.buckets) >= 2
    assert all(v.data_ptr() % 16 == 0 for v in red.views.values())  # 16-byte slots (the fused update's vector loads)
    got = grads(student, teacher, rank, red)
    in_place = sum(1 for n, p in student.named_parameters() if p.grad is not None and p.grad.data_ptr() == red.views[id(p)].data_ptr())
    assert in_place == len(got), (in_place, len(got))  # every gradient lives in its bucket slot
    red.close()
    ref_student, ref_teacher = nano_pair()
    ref_student.ragged_multi_crop = ragged
    acc = None
    for r in range(world):
        g = grads(ref_student, ref_teacher, r, None)
        acc = g if acc is None else {n: acc[n] + g[n] for n in g}
    ok = set(got) == set(acc) and all(torch.allclose(got[n], acc[n] / world, rtol=2e-4, atol=1e-7) for n in got)
    out[rank] = bool(ok)
    dist.destroy_process_group()


def _nano_step_pergroup_worker(rank, world, port, out):
    _nano_step_worker(rank, world, port, out, ragged=False)


def _nano_view_step_worker(rank, world, port, out, ragged=True):
    """BASELINE config 1 in miniature on two ranks: use_dense_prediction=False + DINOLoss.  The view-level Swin forward runs one
    backbone pass per resolution group whatever `ragged_multi_crop` says (swin_transformer.py:729-751), so every block parameter
    gets two gradient contributions per backward while the reducer overlaps (the round-2 advisor's reproduction: with the slot
    handed out twice, 37 of 126 gradients were wrong); also with `ragged_multi_crop` switched off AFTER the reducer exists"""
    _init(rank, world, port)
    import esvit_amd.functional as Fn
    import esvit_amd.loss as L
    import esvit_amd.params as P
    from esvit_amd.engine import GradBucketReducer
    from oracle import ops_ref
    from tests import golden_utils as GU
    from tests.test_composition_cpu import build_nano_view
    for mod in (Fn, L, P):
        mod.ops = ops_ref
    ops_ref.set_act_dtype(torch.float32)
    K = GU.NANO_HEAD["out_dim"]

    def pair():
        student, teacher = build_nano_view(), build_nano_view(teacher=True)
        GU.fill_state_dict(student.state_dict(), 0)
        GU.fill_state_dict(teacher.state_dict(), 7)
        student.head.last_layer.weight_g.data.fill_(1)
        for p in teacher.parameters():
            p.requires_grad = False
        return student, teacher

    def grads(student, teacher, r, reducer):
        crops = GU.make_crops(1, n_local=2, seed=500 + r)
        loss_fn = L.DINOLoss(K, 4, 0.04, 0.04, 0, 1)
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
        return {n: p.grad.detach().clone() for n, p in student.named_parameters() if p.grad is not None}

    student, teacher = pair()
    red = GradBucketReducer(student, bucket_mb=0.25)
    assert red.enabled and red.overlap and len(red.buckets) >= 2
    student.ragged_multi_crop = ragged  # (set after the reducer was built, as the odd-feature-map error message suggests)
    got = grads(student, teacher, rank, red)
    got = grads(student, teacher, rank, red)  # a second step: slots are handed out afresh
    assert all(p.grad.data_ptr() == red.views[id(p)].data_ptr() for p in student.parameters() if p.grad is not None)
    red.close()
    ref_student, ref_teacher = pair()
    ref_student.ragged_multi_crop = ragged
    acc = None
    for r in range(world):
        g = grads(ref_student, ref_teacher, r, None)
        acc = g if acc is None else {n: acc[n] + g[n] for n in g}
    ok = set(got) == set(acc) and all(torch.allclose(got[n], acc[n] / world, rtol=2e-4, atol=1e-7) for n in got)
    out[rank] = bool(ok)
    dist.destroy_process_group()


def _nano_view_step_noragged_worker(rank, world, port, out):
    _nano_view_step_worker(rank, world, port, out, ragged=False)


def _nano_vit_step_worker(rank, world, port, out, ragged=True):
    """the monolithic ViT on two ranks.  ragged: all crops through one set of row-wise launches, one gradient contribution per
    parameter, written straight into the reducer's bucket slots; not ragged: one backbone pass per resolution group
    (vision_transformer.py:186-233), two contributions per parameter (the first into the slot, the sum packed by the hook).  Either
    way the reducer overlaps and the averaged gradients equal the mean of the per-rank gradients"""
    _init(rank, world, port)
    import esvit_amd.functional as Fn
    import esvit_amd.loss as L
    import esvit_amd.params as P
    from esvit_amd.engine import GradBucketReducer
    from oracle import ops_ref
    from tests import golden_utils as GU
    from tests.test_vit_cpu import nano_vit_pair
    for mod in (Fn, L, P):
        mod.ops = ops_ref
    ops_ref.set_act_dtype(torch.float32)
    K = GU.NANO_HEAD["out_dim"]

    def grads(student, teacher, r, reducer):
        crops = GU.make_crops(1, n_local=3, sizes=GU.NANO_VIT["sizes"], seed=400 + r)
        loss_fn = L.DDINOLoss(K, 5, 0.04, 0.04, 0, 1)
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
        return {n: p.grad.detach().clone() for n, p in student.named_parameters() if p.grad is not None}

    student, teacher = nano_vit_pair()
    student.ragged_multi_crop = ragged
    red = GradBucketReducer(student, bucket_mb=0.1)  # as EsvitTrainer arms it
    assert red.enabled and red.overlap and len(red.buckets) >= 2
    got = grads(student, teacher, rank, red)
    # every gradient was produced in (or packed into) its bucket slot
    assert all(p.grad.data_ptr() == red.views[id(p)].data_ptr() for p in student.parameters() if p.grad is not None)
    red.close()
    ref_student, ref_teacher = nano_vit_pair()
    ref_student.ragged_multi_crop = ragged
    acc = None
    for r in range(world):
        g = grads(ref_student, ref_teacher, r, None)
        acc = g if acc is None else {n: acc[n] + g[n] for n in g}
    ok = set(got) == set(acc) and all(torch.allclose(got[n], acc[n] / world, rtol=2e-4, atol=1e-7) for n in got)
    out[rank] = bool(ok)
    dist.destroy_process_group()


def _nano_vit_step_pergroup_worker(rank, world, port, out):
    _nano_vit_step_worker(rank, world, port, out, ragged=False)


def _center_worker(rank, world, port, out):
    _init(rank, world, port)
    import esvit_amd.loss as L
    from oracle import ops_ref
    L.ops = ops_ref  # CPU restatement of the kernels (test infrastructure); the collective logic under test is in loss.py
    ops_ref.set_act_dtype(torch.float32)
    K, B = 256, 3
    g = torch.Generator().manual_seed(7)
    t_cls, t_reg = torch.randn(world, 2 * B, K, generator=g), torch.randn(world, 2 * B * 49, K, generator=g)
    loss = L.DDINOLoss(K, 10, 0.04, 0.04, 0, 1)
    loss.update_center(t_cls[rank], t_reg[rank])   # asynchronous all-reduce ...
    sd = loss.state_dict()                           # ... applied when the centres are next read
    assert sd["center"] is loss.center or torch.equal(sd["center"], loss.center)
    want_c = 0.1 * t_cls.reshape(-1, K).mean(0, keepdim=True)
    want_g = 0.1 * t_reg.reshape(-1, K).mean(0, keepdim=True)
    out[rank] = bool(torch.allclose(loss.center, want_c, atol=1e-6) and torch.allclose(loss.center_grid, want_g, atol=1e-6))
    dist.destroy_process_group()


def _syncbn_worker(rank, world, port, out):
    """CvT attention block (depthwise conv + BatchNorm): batch statistics and the BatchNorm backward reductions all-reduced over the
    ranks give, per rank, exactly the outputs / input gradients of one process that sees the whole batch (SyncBatchNorm semantics,
    main_esvit.py:365-379), and the summed parameter gradients equal the single-process ones."""
    _init(rank, world, port)
    import esvit_amd.functional as Fn
    import esvit_amd.params as P
    from oracle import ops_ref
    Fn.ops = ops_ref
    P.ops = ops_ref
    ops_ref.set_act_dtype(torch.float32)
    C, H, W, nH, B = 64, 6, 6, 1, 2
    g = torch.Generator().manual_seed(11)
    x_all = torch.randn(world * B, H * W, C, generator=g)
    gy_all = torch.randn(world * B, H * W, C, generator=g)
    prm = [1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g), 0.3 * torch.randn(C, 1, 3, 3, generator=g),
           1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g), 0.05 * torch.randn(3 * C, C, 1, 1, generator=g),
           0.02 * torch.randn(3 * C, generator=g), 0.05 * torch.randn(C, C, 1, 1, generator=g), 0.02 * torch.randn(C, generator=g)]

    def run(x, gy, group):
        ps = [p.clone().requires_grad_(True) for p in prm]
        xx = x.clone().requires_grad_(True)
        y = Fn.CvtAttnFn.apply(xx, H, W, nH, 7, None, {"group": group}, *ps)
        y.backward(gy)
        return y.detach(), xx.grad, [p.grad for p in ps]
    y, gx, gp = run(x_all[rank * B:(rank + 1) * B], gy_all[rank * B:(rank + 1) * B], None)      # synchronised statistics
    y1, gx1, gp1 = run(x_all, gy_all, False)                                                      # one process, whole batch
    ok = torch.allclose(y, y1[rank * B:(rank + 1) * B], rtol=1e-4, atol=1e-5) and torch.allclose(gx, gx1[rank * B:(rank + 1) * B], rtol=1e-4, atol=1e-5)
    for a, b in zip(gp, gp1):
        tot = a.clone()
        dist.all_reduce(tot)
        ok = ok and torch.allclose(tot, b, rtol=2e-4, atol=1e-5)
    out[rank] = bool(ok)
    dist.destroy_process_group()


def _syncbn_stem_worker(rank, world, port, out):
    """the residual stem's Conv 3x3 -> BatchNorm -> ReLU unit (functional.ConvBnReluFn, RES_STEM) and the N-layer BatchNorm head
    (functional.DinoHeadBnNFn): statistics and backward reductions synchronised over two ranks == one process on the whole batch"""
    _init(rank, world, port)
    import esvit_amd.functional as Fn
    import esvit_amd.params as P
    from oracle import ops_ref
    Fn.ops = ops_ref
    P.ops = ops_ref
    ops_ref.set_act_dtype(torch.float32)
    g = torch.Generator().manual_seed(17)
    B, Cin, E, H = 2, 8, 16, 6
    x_all = torch.randn(world * B * H * H, Cin, generator=g)
    gy_all = torch.randn(world * B * 3 * 3, E, generator=g)
    prm = [0.2 * torch.randn(E, Cin, 3, 3, generator=g), 1 + 0.1 * torch.randn(E, generator=g), 0.1 * torch.randn(E, generator=g)]

    def run(x, gy, group, nB):
        ps = [p.clone().requires_grad_(True) for p in prm]
        xx = x.clone().requires_grad_(True)
        y = Fn.ConvBnReluFn.apply(xx, (False, nB, H, H, Cin, 3, 2, 1), {"group": group}, *ps)
        y.backward(gy)
        return y.detach(), xx.grad, [p.grad for p in ps]
    rows_in, rows_out = B * H * H, B * 9
    y, gx, gp = run(x_all[rank * rows_in:(rank + 1) * rows_in], gy_all[rank * rows_out:(rank + 1) * rows_out], None, B)
    y1, gx1, gp1 = run(x_all, gy_all, False, world * B)
    ok = torch.allclose(y, y1[rank * rows_out:(rank + 1) * rows_out], rtol=1e-4, atol=1e-5)
    ok = ok and torch.allclose(gx, gx1[rank * rows_in:(rank + 1) * rows_in], rtol=1e-4, atol=1e-5)
    for a, b in zip(gp, gp1):
        tot = a.clone()
        dist.all_reduce(tot)
        ok = ok and torch.allclose(tot, b, rtol=2e-4, atol=1e-5)
    # the BatchNorm head with four layers
    import esvit_amd
    rows, D = 6, 12
    xh = torch.randn(world * rows, D, generator=g)
    gh = torch.randn(world * rows, 32, generator=g)

    def run_head(x, gy, group):
        head = esvit_amd.DINOHead(D, 32, use_bn=True, nlayers=4, hidden_dim=16, bottleneck_dim=8, norm_last_layer=False)
        from tests import golden_utils as GU
        GU.fill_bn_head_n(head.state_dict(), 5)
        head.sync_bn_group = group
        head.train()
        xx = x.clone().requires_grad_(True)
        yy = head(xx)
        yy.backward(gy)
        return yy.detach(), xx.grad, [p.grad for p in head.parameters()]
    y, gx, gp = run_head(xh[rank * rows:(rank + 1) * rows], gh[rank * rows:(rank + 1) * rows], None)
    y1, gx1, gp1 = run_head(xh, gh, False)
    ok = ok and torch.allclose(y, y1[rank * rows:(rank + 1) * rows], rtol=1e-4, atol=1e-5) and torch.allclose(gx, gx1[rank * rows:(rank + 1) * rows], rtol=1e-4, atol=1e-5)
    for a, b in zip(gp, gp1):
        tot = a.clone()
        dist.all_reduce(tot)
        ok = ok and torch.allclose(tot, b, rtol=2e-4, atol=2e-5)
    out[rank] = bool(ok)
    dist.destroy_process_group()


def _extract_worker(rank, world, port, out):
    """eval_knn.py:165-190 over two ranks: each rank runs its share of the batches, rank 0 ends up with every row at its
    dataset index"""
    _init(rank, world, port)
    import esvit_amd.functional as Fn
    import esvit_amd.params as P
    from esvit_amd import eval as E
    from oracle import ops_ref
    from tests.test_composition_cpu import IndexedSet, build_nano_backbone
    Fn.ops = ops_ref
    P.ops = ops_ref
    ops_ref.set_act_dtype(torch.float32)
    model = build_nano_backbone()
    x = torch.randn(12, 3, 64, 64, generator=torch.Generator().manual_seed(3))
    mine = list(range(rank, 12, world))  # DistributedSampler-style interleaving
    loader = torch.utils.data.DataLoader(torch.utils.data.Subset(IndexedSet(x), mine), batch_size=3)
    full = torch.utils.data.DataLoader(IndexedSet(x), batch_size=3)

    class Both:  # len(dataset) must be the FULL set (eval_knn.py:175), batches are this rank's share
        dataset = full.dataset

        def __iter__(self):
            return iter(loader)
    feats = E.extract_features(model, Both(), use_cuda=False)
    if rank == 0:
        with torch.no_grad():
            want = model(x)
        out[rank] = bool(feats.shape == want.shape and torch.allclose(feats, want, rtol=1e-5, atol=1e-6))
    else:
        out[rank] = feats is None
    dist.destroy_process_group()


@pytest.mark.parametrize("worker,port", [(_reducer_worker, 29611), (_center_worker, 29612), (_syncbn_worker, 29613), (_extract_worker, 29614),
                                         (_nano_step_worker, 29615), (_nano_step_pergroup_worker, 29616), (_nano_vit_step_worker, 29617), (_nano_vit_step_pergroup_worker, 29618),
                                         (_nano_view_step_worker, 29619), (_nano_view_step_noragged_worker, 29620), (_reducer_bf16_worker, 29621), (_syncbn_stem_worker, 29622)])
def test_world2_gloo(worker, port, lib_built):
    world = 2
    out = mp.Manager().dict()
    mp.spawn(worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}, dict(out)
