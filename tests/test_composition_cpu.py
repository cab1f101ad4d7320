"""CPU test of the HOST-SIDE composition (autograd glue, manual backward formulas, index-map plumbing, loss
tables) of the product modules: esvit_amd.ops is monkeypatched with the plain-PyTorch op restatement
(oracle/ops_ref.py) so the whole EsViT step can run without a GPU and be compared with the golden vectors
generated from the reference.  The HIP kernels themselves are checked op-by-op in tests/test_kernels_gpu.py."""
import os

import pytest
import torch

from oracle import esvit_oracle as O
from oracle import ops_ref
from oracle import ref_loader as RL
from tests import golden_utils as GU
from tests.test_oracle_cpu import GOLD, probe_close


@pytest.fixture()
def cpu_ops(monkeypatch, lib_built):
    import esvit_amd.functional as Fn
    import esvit_amd.loss as L
    import esvit_amd.params as P
    for mod in (Fn, L, P):
        monkeypatch.setattr(mod, "ops", ops_ref)
    ops_ref.set_act_dtype(torch.float32)
    P.clear()
    Fn._GEOM.clear()
    yield ops_ref
    ops_ref.set_act_dtype(torch.float32)
    P.clear()
    Fn._GEOM.clear()


def build_nano(teacher=False, window=None):
    from esvit_amd import models
    cfg = RL.swin_config(embed_dim=GU.NANO["embed_dim"], depths=GU.NANO["depths"], heads=GU.NANO["heads"],
                         window=window or GU.NANO["window"])
    m = models.build_model(cfg, is_teacher=teacher, use_dense_prediction=True)
    hk = dict(hidden_dim=GU.NANO_HEAD["hidden_dim"], bottleneck_dim=GU.NANO_HEAD["bottleneck_dim"])
    m.head = models.DINOHead(m.num_features, GU.NANO_HEAD["out_dim"], norm_last_layer=True, **hk)
    m.head_dense = models.DINOHead(m.num_features, GU.NANO_HEAD["out_dim"], norm_last_layer=False, **hk)
    return m


def build_nano_view(teacher=False):
    """BASELINE config 1 in miniature: use_dense_prediction=False, view-level head only (main_esvit.py:235-254 without
    --use_dense_prediction)"""
    from esvit_amd import models
    cfg = RL.swin_config(embed_dim=GU.NANO["embed_dim"], depths=GU.NANO["depths"], heads=GU.NANO["heads"], window=GU.NANO["window"])
    m = models.build_model(cfg, is_teacher=teacher, use_dense_prediction=False)
    hk = dict(hidden_dim=GU.NANO_HEAD["hidden_dim"], bottleneck_dim=GU.NANO_HEAD["bottleneck_dim"])
    m.head = models.DINOHead(m.num_features, GU.NANO_HEAD["out_dim"], norm_last_layer=True, **hk)
    return m


def nano_pair(window=None):
    student, teacher = build_nano(window=window), build_nano(teacher=True, window=window)
    GU.fill_state_dict(student.state_dict(), 0)
    GU.fill_state_dict(teacher.state_dict(), 7)
    student.head.last_layer.weight_g.data.fill_(1)
    for p in teacher.parameters():
        p.requires_grad = False
    return student, teacher


def run_nano_step(nano, student, teacher, loss_mod, crops, dev="cpu"):
    K = GU.NANO_HEAD["out_dim"]
    loss_fn = loss_mod.DDINOLoss(K, 10, 0.04, 0.07, 5, 10).to(dev)
    loss_fn.center.copy_(nano["center0"])
    loss_fn.center_grid.copy_(nano["center_grid0"])
    t_out = teacher(crops[:2])
    s_out = student(crops)
    loss = loss_fn(s_out, t_out, 2, None)
    loss.backward()
    return s_out, t_out, loss, loss_fn


def check_ragged_equals_reference_schedule(loss_mod, dev="cpu", tol=1e-5):
    """the ragged multi-crop route (all resolution groups as one row matrix) and the reference's schedule (one backbone pass
    per group, swin_transformer.py:729-751) give the same outputs, loss and parameter gradients"""
    nano = torch.load(os.path.join(GOLD, "nano_step.pt"), weights_only=False)
    crops = [c.to(dev) for c in GU.make_crops(2)]
    res = []
    for ragged in (True, False):
        student, teacher = nano_pair()
        student, teacher = student.to(dev), teacher.to(dev)
        student.ragged_multi_crop = ragged
        s_out, _, loss, _ = run_nano_step(nano, student, teacher, loss_mod, crops, dev)
        res.append((s_out, loss, {n: p.grad.detach().clone() for n, p in student.named_parameters() if p.grad is not None}))
    (sa, la, ga), (sb, lb, gb) = res
    for a, b in zip(sa[:3], sb[:3]):
        assert (a - b).abs().max().item() <= tol * (b.abs().max().item() + 1e-12)
    assert abs(la.item() - lb.item()) <= tol
    assert ga.keys() == gb.keys()
    for n in ga:
        assert (ga[n] - gb[n]).abs().max().item() <= 20 * tol * (gb[n].abs().max().item() + 1e-12), n


def test_ragged_multi_crop_equals_reference_schedule(cpu_ops):
    import esvit_amd.loss as L
    check_ragged_equals_reference_schedule(L)


def test_one_group_of_several_crops_rides_the_ragged_route(cpu_ops):
    """the teacher's input -- two crops of ONE resolution -- is read where it lies (no torch.cat pass over the images): same outputs as
    the reference's schedule, which concatenates them first (swin_transformer.py:741)"""
    crops = [torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(7)),
             torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(8))]
    outs, routes = [], []
    for ragged in (True, False):
        teacher = build_nano()
        GU.fill_state_dict(teacher.state_dict(), 0)
        teacher.ragged_multi_crop = ragged
        calls = []
        orig = teacher.forward_feature_maps_multi
        teacher.forward_feature_maps_multi = lambda groups, _o=orig, _c=calls: (_c.append(len(groups)), _o(groups))[1]
        with torch.no_grad():
            outs.append(teacher(crops))
        routes.append(calls)
    assert routes == [[1], []]
    for a, b in zip(outs[0][:3], outs[1][:3]):
        assert a.shape == b.shape and (a - b).abs().max().item() <= 1e-5 * (b.abs().max().item() + 1e-12)


def test_odd_feature_maps_take_the_per_group_schedule(cpu_ops):
    """crops whose feature map is odd at a PatchMerging (112^2 at four stages: 28, 14, 7) cannot ride the ragged route (it has no
    padding step): the default forward falls back to the reference's per-group schedule, which pads them (swin_transformer.py:406-408),
    instead of refusing -- same outputs as with the ragged route switched off by hand"""
    crops = [torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(5)),
             torch.randn(2, 3, 112, 112, generator=torch.Generator().manual_seed(6))]
    outs = []
    for ragged in (True, False):
        student = build_nano()
        GU.fill_state_dict(student.state_dict(), 0)
        student.ragged_multi_crop = ragged
        assert not student._even_maps([224, 112]) and student._even_maps([224, 96])
        with torch.no_grad():
            outs.append(student(crops))
    for a, b in zip(outs[0][:3], outs[1][:3]):
        assert torch.equal(a, b)
    assert outs[0][3] == outs[1][3] == [49, 16]


def check_odd_batches_vs_oracle(loss_mod, dev="cpu", window=None, tol=2e-4, gtol=5e-3):
    """per-GPU batches that are not multiples of anything (B = 1, 3): outputs, loss and gradient norms of the nano model vs the
    CPU oracle on the same weights -- exercises the split-K / workgroup sizing and the ragged row matrix away from round sizes"""
    from oracle import esvit_oracle as O
    cfg = dict(GU.NANO14 if window == 14 else GU.NANO)
    K = GU.NANO_HEAD["out_dim"]
    for B in (1, 3):
        student, teacher = nano_pair(window=window)
        sd = {k: v.clone() for k, v in student.state_dict().items()}
        tsd = {k: v.clone() for k, v in teacher.state_dict().items()}
        crops = GU.make_crops(B, seed=40 + B)
        names = [n for n, p in student.named_parameters() if p.requires_grad]
        leaf = {n: sd[n].clone().requires_grad_(True) for n in names}
        full = dict(sd)
        full.update(leaf)
        s_ref = O.swin_multicrop(full, crops, cfg)
        with torch.no_grad():
            t_ref = O.swin_multicrop(tsd, crops[:2], cfg)
        c0 = torch.zeros(1, K)
        l_ref, _, _ = O.ddino_loss(s_ref, t_ref, c0, c0, O.teacher_temp(0, 0.04, 0.04, 0, 1), 10)
        l_ref.backward()
        student, teacher = student.to(dev), teacher.to(dev)
        loss_fn = loss_mod.DDINOLoss(K, 10, 0.04, 0.04, 0, 1).to(dev)
        dcrops = [c.to(dev) for c in crops]
        t_out = teacher(dcrops[:2])
        s_out = student(dcrops)
        loss = loss_fn(s_out, t_out, 0, None)
        loss.backward()
        for a, b in zip(s_out[:3], s_ref[:3]):
            assert ((a.float().cpu() - b).abs().max() / (b.abs().max() + 1e-12)).item() < tol, B
        assert abs(loss.item() - l_ref.item()) < tol, (B, loss.item(), l_ref.item())
        for n, p in student.named_parameters():
            if p.requires_grad:
                ref = leaf[n].grad.norm().item()
                assert abs(p.grad.norm().item() - ref) <= gtol * (ref + 1e-6), (B, n, p.grad.norm().item(), ref)


def test_odd_batches_match_oracle(cpu_ops):
    import esvit_amd.loss as L
    check_odd_batches_vs_oracle(L)


def check_drop_path_vs_oracle(loss_mod, dev="cpu", tol=2e-4, gtol=5e-3):
    """stochastic depth (DROP_PATH_RATE 0.4, training mode) with the per-sample keep factors FIXED by the test and handed to both
    sides: the per-row DropPath scale of the ragged route -- in the GEMM epilogues, the LayerNorm-backward casts and the
    gradient the next block emits for its predecessor (Fn.SwinBlockMultiFn's shadow output) -- vs the CPU oracle"""
    from esvit_amd import models
    from esvit_amd.models.swin_transformer import DropPath
    from oracle import esvit_oracle as O
    K = GU.NANO_HEAD["out_dim"]
    B = 2
    cfg = RL.swin_config(embed_dim=GU.NANO["embed_dim"], depths=GU.NANO["depths"], heads=GU.NANO["heads"], window=GU.NANO["window"], drop_path=0.4)
    student = models.build_model(cfg, is_teacher=False, use_dense_prediction=True)
    hk = dict(hidden_dim=GU.NANO_HEAD["hidden_dim"], bottleneck_dim=GU.NANO_HEAD["bottleneck_dim"])
    student.head = models.DINOHead(student.num_features, K, norm_last_layer=True, **hk)
    student.head_dense = models.DINOHead(student.num_features, K, norm_last_layer=False, **hk)
    _, teacher = nano_pair()
    GU.fill_state_dict(student.state_dict(), 0)
    student.head.last_layer.weight_g.data.fill_(1)
    student.train()
    sd = {k: v.clone() for k, v in student.state_dict().items()}
    tsd = {k: v.clone() for k, v in teacher.state_dict().items()}
    crops = GU.make_crops(B, seed=77)
    group_sizes = [2 * B, 8 * B]  # samples of the 224^2 group, then of the 96^2 group (row order of the ragged matrix)
    nS = sum(group_sizes)
    gen = torch.Generator().manual_seed(5)
    blocks = [b for layer in student.layers for b in layer.blocks]
    factors = []
    for b in blocks:
        p = b.drop_path.drop_prob if isinstance(b.drop_path, DropPath) else 0.0
        keep = 1.0 - (p or 0.0)
        f = [(keep + torch.rand(nS, generator=gen)).floor_().div_(keep) for _ in range(2)]
        factors.append(f)
    assert any((f[0] == 0).any() for f in factors) and any((f[1] == 0).any() for f in factors)  # some samples are dropped
    # oracle: per group, per block (attention branch, MLP branch) factors of that group's samples
    drop_scales = []
    off = 0
    for n in group_sizes:
        drop_scales.append([(f[0][off:off + n], f[1][off:off + n]) for f in factors])
        off += n
    names = [n for n, p in student.named_parameters() if p.requires_grad]
    leaf = {n: sd[n].clone().requires_grad_(True) for n in names}
    full = dict(sd)
    full.update(leaf)
    s_ref = O.swin_multicrop(full, crops, dict(GU.NANO), drop_scales=drop_scales)
    with torch.no_grad():
        t_ref = O.swin_multicrop(tsd, crops[:2], dict(GU.NANO))
    c0 = torch.zeros(1, K)
    l_ref, _, _ = O.ddino_loss(s_ref, t_ref, c0, c0, O.teacher_temp(0, 0.04, 0.04, 0, 1), 10)
    l_ref.backward()
    # ours: same factors installed where SwinTransformer._draw_drop_path would put its own draw
    student, teacher = student.to(dev), teacher.to(dev)
    student._draw_drop_path = lambda nB, device: None
    for b, f in zip(blocks, factors):
        if isinstance(b.drop_path, DropPath) and b.drop_path.drop_prob:
            b.__dict__["_dp_pending"] = (f[0].to(dev), f[1].to(dev))
    loss_fn = loss_mod.DDINOLoss(K, 10, 0.04, 0.04, 0, 1).to(dev)
    dcrops = [c.to(dev) for c in crops]
    with torch.no_grad():
        t_out = teacher(dcrops[:2])
    s_out = student(dcrops)
    loss = loss_fn(s_out, t_out, 0, None)
    loss.backward()
    for a, b in zip(s_out[:3], s_ref[:3]):
        assert ((a.float().cpu() - b).abs().max() / (b.abs().max() + 1e-12)).item() < tol
    assert abs(loss.item() - l_ref.item()) < tol, (loss.item(), l_ref.item())
    for n, p in student.named_parameters():
        if p.requires_grad:
            ref = leaf[n].grad
            err = (p.grad.float().cpu() - ref).abs().max().item()
            assert err <= gtol * (ref.abs().max().item() + 1e-7), (n, err, ref.abs().max().item())


def test_drop_path_matches_oracle(cpu_ops):
    import esvit_amd.loss as L
    check_drop_path_vs_oracle(L)


def test_state_dict_layout_matches_golden():
    nano = torch.load(os.path.join(GOLD, "nano_step.pt"), weights_only=False)
    m = build_nano()
    assert [(k, tuple(v.shape), str(v.dtype)) for k, v in m.state_dict().items()] == nano["keys"]
    assert [n for n, _ in m.named_parameters()] == nano["param_names"]
    assert [n for n, p in m.named_parameters() if p.requires_grad] == nano["trainable"]


def test_composition_fp32_matches_reference_golden(cpu_ops):
    import esvit_amd.loss as L
    nano = torch.load(os.path.join(GOLD, "nano_step.pt"), weights_only=False)
    student, teacher = nano_pair()
    crops = GU.make_crops(2)
    s_out, t_out, loss, loss_fn = run_nano_step(nano, student, teacher, L, crops)
    for nm, t in (("s_cls", s_out[0]), ("s_reg", s_out[1]), ("s_fea", s_out[2]), ("t_cls", t_out[0]), ("t_reg", t_out[1]),
                  ("t_fea", t_out[2])):
        probe_close(nm, t, nano[nm])
    assert (list(s_out[3]), list(t_out[3])) == nano["npatch"]
    assert abs(loss.item() - nano["ddino_loss"]) < 2e-5
    assert (loss_fn.center - nano["center1"]).abs().max().item() < 1e-6
    assert (loss_fn.center_grid - nano["center_grid1"]).abs().max().item() < 1e-6
    assert [n for n, p in student.named_parameters() if p.grad is None] == nano["no_grad"]
    for n, p in student.named_parameters():
        if p.grad is not None:
            probe_close("grad " + n, p.grad, nano["grads"][n], rtol=1e-3)
    # second call sees the updated centres
    with torch.no_grad():
        l2 = loss_fn([t.detach() if torch.is_tensor(t) else t for t in s_out], t_out, 2, None)
    assert abs(l2.item() - nano["ddino_loss_2"]) < 2e-5
    # view-level DINOLoss on the two global crops
    vl = L.DINOLoss(GU.NANO_HEAD["out_dim"], 2, 0.04, 0.07, 5, 10)
    vl.center.copy_(nano["center0"])
    with torch.no_grad():
        cls2 = student.head(student.forward_features(torch.cat(crops[:2]))[0])
        lv = vl(cls2, t_out[0], 2, None)
    assert abs(lv.item() - nano["dino_loss_2crops"]) < 2e-5
    assert (vl.center - nano["dino_center1"]).abs().max().item() < 1e-6
    with torch.no_grad():
        probe_close("last_attn", student.forward_selfattention(crops[0]), nano["last_attn"])


def test_config1_view_only_composition_matches_reference_golden(cpu_ops):
    """BASELINE config 1 host logic: use_dense_prediction=False forward (swin_transformer.py:753-763) + DINOLoss + backward"""
    import esvit_amd.loss as L
    nano = torch.load(os.path.join(GOLD, "nano_step.pt"), weights_only=False)
    student, teacher = build_nano_view(), build_nano_view(teacher=True)
    GU.fill_state_dict(student.state_dict(), 0)
    GU.fill_state_dict(teacher.state_dict(), 7)
    student.head.last_layer.weight_g.data.fill_(1)
    crops = GU.make_crops(2)[:2]
    with torch.no_grad():
        t_out = teacher(crops)
    s_out = student(crops)
    assert torch.is_tensor(s_out) and s_out.shape == (4, GU.NANO_HEAD["out_dim"])
    probe_close("t_cls", t_out, nano["t_cls"])
    vl = L.DINOLoss(GU.NANO_HEAD["out_dim"], 2, 0.04, 0.07, 5, 10)
    vl.center.copy_(nano["center0"])
    loss = vl(s_out, t_out, 2, None)
    loss.backward()
    assert abs(loss.item() - nano["dino_loss_2crops"]) < 2e-5
    assert (vl.center - nano["dino_center1"]).abs().max().item() < 1e-6
    assert [n for n, p in student.named_parameters() if p.requires_grad and p.grad is None] == []
    # the same loss through the oracle restatement
    sd = {k: v.clone() for k, v in student.state_dict().items()}
    l_o, _ = O.dino_loss(O.swin_multicrop(sd, crops, dict(GU.NANO), dense=False), t_out, nano["center0"], O.teacher_temp(2, 0.04, 0.07, 5, 10), 2)
    assert abs(l_o.item() - nano["dino_loss_2crops"]) < 2e-5


def test_swin_forward_return_n_last_blocks_matches_reference_golden(cpu_ops):
    """eval_linear.py's feature hook (swin_transformer.py:799-837): host logic vs the reference golden"""
    nano = torch.load(os.path.join(GOLD, "nano_step.pt"), weights_only=False)
    student = build_nano()
    GU.fill_state_dict(student.state_dict(), 0)
    student.eval()
    crops = GU.make_crops(2)
    depth = list(GU.NANO["depths"])
    with torch.no_grad():
        f3 = student.forward_return_n_last_blocks(crops[0], n=3, depth=depth)
        f1 = student.forward_return_n_last_blocks(crops[2], n=1, depth=depth)
    assert torch.allclose(f3, nano["last_blocks_n3"], rtol=2e-4, atol=1e-5)
    assert torch.allclose(f1, nano["last_blocks_n1_local"], rtol=2e-4, atol=1e-5)


def test_composition_bf16_emulation_error_budget(cpu_ops):
    """bf16 activation storage emulated on CPU: records how far bf16 rounding alone moves the loss/gradients, which
    is the tolerance the -m gpu bf16 tests are allowed (the HIP kernels round at the same points)."""
    import esvit_amd.loss as L
    nano = torch.load(os.path.join(GOLD, "nano_step.pt"), weights_only=False)
    ops_ref.set_act_dtype(torch.bfloat16)
    student, teacher = nano_pair()
    s_out, t_out, loss, _ = run_nano_step(nano, student, teacher, L, GU.make_crops(2))
    assert abs(loss.item() - nano["ddino_loss"]) < 2e-2
    worst = 0.0
    for n, p in student.named_parameters():
        if p.grad is not None:
            ref = nano["grad_norms"][n]
            worst = max(worst, abs(p.grad.norm().item() - ref) / (ref + 1e-12))
    assert worst < 0.15, worst


def run_nano14_step(student, teacher, loss_mod, dev="cpu"):
    crops = [c.to(dev) for c in GU.make_crops(1, n_local=2)]
    loss_fn = loss_mod.DDINOLoss(GU.NANO_HEAD["out_dim"], 4, 0.04, 0.07, 5, 10).to(dev)
    t_out = teacher(crops[:2])
    s_out = student(crops)
    loss = loss_fn(s_out, t_out, 2, None)
    loss.backward()
    return s_out, t_out, loss


def check_nano14(g14, student, s_out, t_out, loss, rt, loss_tol, grad_tol, probes=True):
    for nm, t in (("s_cls", s_out[0]), ("s_reg", s_out[1]), ("s_fea", s_out[2]), ("t_cls", t_out[0]), ("t_reg", t_out[1])):
        probe_close(nm, t.float().cpu(), g14[nm], rtol=rt)
    assert abs(loss.item() - g14["ddino_loss"]) < loss_tol, (loss.item(), g14["ddino_loss"])
    worst = 0.0
    for n, p in student.named_parameters():
        if p.grad is not None:
            ref = g14["grad_norms"][n]
            worst = max(worst, abs(p.grad.norm().item() - ref) / (ref + 1e-12))
            if probes and n in g14["grads"]:
                probe_close("grad " + n, p.grad.cpu(), g14["grads"][n], rtol=2e-3)
    assert worst < grad_tol, worst


def test_composition_w14_matches_reference_golden(cpu_ops):
    import esvit_amd.loss as L
    g14 = torch.load(os.path.join(GOLD, "nano14_step.pt"), weights_only=False)
    student, teacher = nano_pair(window=14)
    assert [(k, tuple(v.shape), str(v.dtype)) for k, v in student.state_dict().items()] == g14["keys"]
    s_out, t_out, loss = run_nano14_step(student, teacher, L)
    check_nano14(g14, student, s_out, t_out, loss, rt=2e-4, loss_tol=2e-5, grad_tol=1e-3)


# ---- CvT (BASELINE config 5) -------------------------------------------------------------------
def build_nano_cvt(teacher=False):
    from esvit_amd import models
    cfg = RL.cvt_config(dims=GU.NANO_CVT["dims"], heads=GU.NANO_CVT["heads"], depths=GU.NANO_CVT["depths"])
    m = models.build_model(cfg, is_teacher=teacher, use_dense_prediction=True)
    hk = dict(hidden_dim=GU.NANO_HEAD["hidden_dim"], bottleneck_dim=GU.NANO_HEAD["bottleneck_dim"])
    fea = GU.NANO_CVT["dims"][-1]
    m.head = models.DINOHead(fea, GU.NANO_HEAD["out_dim"], norm_last_layer=True, **hk)
    m.head_dense = models.DINOHead(fea, GU.NANO_HEAD["out_dim"], norm_last_layer=False, **hk)
    return m


def nano_cvt_pair(dev="cpu"):
    student, teacher = build_nano_cvt(), build_nano_cvt(teacher=True)
    GU.fill_state_dict(student.state_dict(), 0)
    GU.fill_state_dict(teacher.state_dict(), 7)
    for m in (student, teacher):
        for k, v in m.state_dict().items():
            if k.endswith("running_var"):
                v.abs_().add_(0.5)
    student.head.last_layer.weight_g.data.fill_(1)
    for p in teacher.parameters():
        p.requires_grad = False
    return student.to(dev), teacher.to(dev)


def run_nano_cvt_step(student, teacher, loss_mod, dev="cpu"):
    crops = [c.to(dev) for c in GU.make_crops(2, n_local=3, sizes=GU.NANO_CVT["sizes"])]
    loss_fn = loss_mod.DDINOLoss(GU.NANO_HEAD["out_dim"], 5, 0.04, 0.07, 5, 10).to(dev)
    t_out = teacher(crops[:2])
    s_out = student(crops)
    loss = loss_fn(s_out, t_out, 2, None)
    loss.backward()
    return s_out, t_out, loss


def check_nano_cvt(g, student, s_out, t_out, loss, rt, loss_tol, grad_tol, buf_tol):
    assert list(s_out[3]) == g["npatch"][0] and list(t_out[3]) == g["npatch"][1]
    for nm, t in (("s_cls", s_out[0]), ("s_reg", s_out[1]), ("s_fea", s_out[2]), ("t_cls", t_out[0]), ("t_reg", t_out[1]), ("t_fea", t_out[2])):
        probe_close(nm, t.float().cpu(), g[nm], rtol=rt)
    assert abs(loss.item() - g["ddino_loss"]) < loss_tol, (loss.item(), g["ddino_loss"])
    got = {n: p.grad for n, p in student.named_parameters() if p.grad is not None}
    assert sorted(got) == sorted(g["grad_norms"])
    for n, ref in g["grad_norms"].items():
        assert abs(got[n].norm().item() - ref) <= grad_tol * ref + 1e-9, (n, got[n].norm().item(), ref)
    sd = student.state_dict()
    for k, v in g["bn_buffers"].items():
        assert torch.allclose(sd[k].float().cpu(), v.float(), rtol=buf_tol, atol=buf_tol), k


def test_cvt_state_dict_layout_matches_golden():
    g = torch.load(os.path.join(GOLD, "nano_cvt_step.pt"), weights_only=False)
    student = build_nano_cvt()
    assert [(k, tuple(v.shape), str(v.dtype)) for k, v in student.state_dict().items()] == g["keys"]
    assert [n for n, _ in student.named_parameters()] == g["param_names"]


def test_cvt_composition_matches_reference_golden(cpu_ops):
    """the CvT module tree + autograd glue (ConvEmbedFn / CvtAttnFn / CvtFfnFn) on the torch restatement of every kernel"""
    import esvit_amd.loss as L
    g = torch.load(os.path.join(GOLD, "nano_cvt_step.pt"), weights_only=False)
    student, teacher = nano_cvt_pair()
    s_out, t_out, loss = run_nano_cvt_step(student, teacher, L)
    check_nano_cvt(g, student, s_out, t_out, loss, rt=3e-4, loss_tol=2e-5, grad_tol=2e-3, buf_tol=1e-4)


def test_cvt_eval_mode_matches_reference_golden(cpu_ops):
    """inference consumers (eval_linear.py / eval_knn.py): eval-mode BatchNorm and forward_return_n_last_blocks"""
    g = torch.load(os.path.join(GOLD, "nano_cvt_step.pt"), weights_only=False)
    student, _ = nano_cvt_pair()
    student.eval()
    crops = GU.make_crops(2, n_local=3, sizes=GU.NANO_CVT["sizes"])
    with torch.no_grad():
        cls, region = student.forward_features(crops[0])
        probe_close("eval cls", cls, g["eval_cls"], rtol=3e-4)
        probe_close("eval region", region, g["eval_region"], rtol=3e-4)
        feats = student.forward_return_n_last_blocks(crops[2], n=2, depth=list(GU.NANO_CVT["depths"]))
    assert torch.allclose(feats, g["eval_last_blocks"], rtol=3e-4, atol=1e-5)


def build_cvt_variant(case, teacher=False):
    from esvit_amd import models
    m = models.build_model(RL.cvt_config(**case["cfg"]), is_teacher=teacher, use_dense_prediction=True)
    hk = dict(hidden_dim=GU.NANO_HEAD["hidden_dim"], bottleneck_dim=GU.NANO_HEAD["bottleneck_dim"])
    fea = case["cfg"]["dims"][-1]
    m.head = models.DINOHead(fea, GU.NANO_HEAD["out_dim"], norm_last_layer=True, **hk)
    m.head_dense = models.DINOHead(fea, GU.NANO_HEAD["out_dim"], norm_last_layer=False, **hk)
    return m


def check_cvt_variant(name, loss_mod, dev="cpu", rt=3e-4, loss_tol=2e-5, grad_tol=2e-3, buf_tol=1e-4, probes=True):
    """one training step of a CvT variant (REL_POS_EMBED / SHIFT / RES_STEM) against the reference's own module"""
    case = GU.NANO_CVT_VARIANTS[name]
    g = torch.load(os.path.join(GOLD, "nano_cvt_variants.pt"), weights_only=False)[name]
    student, teacher = build_cvt_variant(case), build_cvt_variant(case, teacher=True)
    assert [(k, tuple(v.shape), str(v.dtype)) for k, v in student.state_dict().items()] == g["keys"]
    assert [n for n, _ in student.named_parameters()] == g["param_names"]
    GU.fill_state_dict(student.state_dict(), 0)
    GU.fill_state_dict(teacher.state_dict(), 7)
    for m in (student, teacher):
        for k, v in m.state_dict().items():
            if k.endswith("running_var"):
                v.abs_().add_(0.5)
    student.head.last_layer.weight_g.data.fill_(1)
    for p in teacher.parameters():
        p.requires_grad = False
    student, teacher = student.to(dev), teacher.to(dev)
    crops = [c.to(dev) for c in GU.make_crops(2, n_local=case["n_local"], sizes=case["sizes"])]
    loss_fn = loss_mod.DDINOLoss(GU.NANO_HEAD["out_dim"], 2 + case["n_local"], 0.04, 0.07, 5, 10).to(dev)
    t_out = teacher(crops[:2])
    s_out = student(crops)
    loss = loss_fn(s_out, t_out, 2, None)
    loss.backward()
    if probes:
        check_nano_cvt(g, student, s_out, t_out, loss, rt, loss_tol, grad_tol, buf_tol)
    else:  # (bf16 on the GPU: loss and gradient norms, like the plain nano CvT step)
        got = {n: p.grad for n, p in student.named_parameters() if p.grad is not None}
        GU.record_parity(test="cvt_variant_" + name, prec="bf16", abs_err=abs(loss.item() - g["ddino_loss"]),
                         worst_grad_norm_rel=max(max(abs(got[n].norm().item() - ref) - 1e-6, 0.0) / (ref + 1e-12) for n, ref in g["grad_norms"].items() if n in got))
        assert abs(loss.item() - g["ddino_loss"]) < loss_tol, (loss.item(), g["ddino_loss"])
        assert sorted(got) == sorted(g["grad_norms"])
        for n, ref in g["grad_norms"].items():
            assert abs(got[n].norm().item() - ref) <= grad_tol * ref + 1e-6, (n, got[n].norm().item(), ref)
    if probes:
        for n, p in student.named_parameters():
            if "rel_pos_bias_table" in n or "stem" in n:
                probe_close(n, p.grad.float().cpu(), g["grads"][n], rtol=max(rt, grad_tol))
    return student


@pytest.mark.parametrize("name", sorted(GU.NANO_CVT_VARIANTS))
def test_cvt_variants_composition_matches_reference_golden(name, cpu_ops):
    """REL_POS_EMBED / SHIFT (s1_rpe.yaml, s1_shift.yaml, s1_rpe_shift.yaml) on the torch restatement of every kernel"""
    import esvit_amd.loss as L
    check_cvt_variant(name, L)


def test_cvt_variants_refuse_what_the_reference_cannot_run(cpu_ops):
    """a map narrower than the window (bias / mask shapes) or, with SHIFT, not a multiple of it fails in the reference as well"""
    m = build_cvt_variant(GU.NANO_CVT_VARIANTS["rpe_shift"])
    with pytest.raises(RuntimeError, match="smaller than"):
        m([torch.randn(1, 3, 24, 24)])
    with pytest.raises(RuntimeError, match="multiple"):
        m([torch.randn(1, 3, 64, 64)])


# ---- eval_knn.py consumers ---------------------------------------------------
class IndexedSet(torch.utils.data.Dataset):
    """the reference's ReturnIndexDataset (eval_knn.py:235-238): (sample, position in the dataset)"""

    def __init__(self, x):
        self.x = x

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        return self.x[i], i


def build_nano_backbone():
    """build_model(config, is_teacher=True) with NUM_CLASSES 0 as eval_knn.py:102 builds it: forward returns the cls features"""
    from esvit_amd import models
    cfg = RL.swin_config(embed_dim=GU.NANO["embed_dim"], depths=GU.NANO["depths"], heads=GU.NANO["heads"], window=GU.NANO["window"])
    m = models.build_model(cfg, is_teacher=True)
    GU.fill_state_dict(m.state_dict(), 21)
    return m.eval()


def check_extract_features(dev="cpu", tol=1e-5):
    from esvit_amd import eval as E
    model = build_nano_backbone().to(dev)
    x = torch.randn(10, 3, 64, 64, generator=torch.Generator().manual_seed(3))
    order = torch.randperm(10, generator=torch.Generator().manual_seed(4)).tolist()
    loader = torch.utils.data.DataLoader(torch.utils.data.Subset(IndexedSet(x), order), batch_size=4)
    feats = E.extract_features(model, loader, use_cuda=(dev != "cpu"))
    with torch.no_grad():
        want = model(x.to(dev))
    assert feats.shape == want.shape == (10, model.num_features)
    assert (feats - want).abs().max().item() <= tol * want.abs().max().item()


def test_extract_features_places_rows_by_index(cpu_ops):
    check_extract_features()


def test_knn_classifier_host_logic_matches_reference_golden(cpu_ops, monkeypatch):
    """esvit_amd.eval.knn_classifier with the GEMM swapped for the CPU restatement: chunking, top-k vote and ranking vs the
    reference's numbers"""
    from esvit_amd import eval as E
    monkeypatch.setattr(E, "ops", cpu_ops)
    gold = torch.load(os.path.join(GOLD, "knn.pt"), weights_only=False)
    for c, want in zip(GU.KNN_CASES, gold["top"]):
        xtr, ytr, xte, yte = GU.make_knn_set(c["seed"], noise=c["noise"])
        got = E.knn_classifier(xtr, ytr, xte, yte, c["k"], c["T"], num_classes=10)
        assert got == pytest.approx(want, abs=1e-9), (c, got, want)


def check_ref_checkpoint(dev="cpu", tol=2e-5, top_tol=1.0):
    """eval_knn.py end to end over a checkpoint FILE the reference wrote (utils.save_on_master; DistributedDataParallel student, plain
    teacher, heads, loss state): load_pretrained_weights -> extract_features -> knn_classifier, against the reference's own features
    and votes for both checkpoint keys"""
    from esvit_amd import eval as E
    from esvit_amd import models
    c = GU.REF_CKPT
    gold = torch.load(os.path.join(GOLD, "ref_checkpoint.pt"), weights_only=False)
    path = os.path.join(GOLD, "ref_checkpoint.pth")
    xtr, ytr, xte, yte = GU.ref_ckpt_data()
    for key in ("teacher", "student"):
        cfg = RL.swin_config(embed_dim=c["embed_dim"], depths=c["depths"], heads=c["heads"], window=c["window"])
        model = models.build_model(cfg, is_teacher=True)
        msg = E.load_pretrained_weights(model, path, key, "swin_nano", 4)
        assert not msg.missing_keys and all(k.startswith(("head.", "head_dense.")) for k in msg.unexpected_keys)
        model = model.to(dev).eval()
        feats = []
        for x in (xtr, xte):
            loader = torch.utils.data.DataLoader(IndexedSet(x), batch_size=32)
            feats.append(E.extract_features(model, loader, use_cuda=(dev != "cpu")))
        ntr, nte = (torch.nn.functional.normalize(f, dim=1, p=2) for f in feats)
        top = E.knn_classifier(ntr, ytr.to(ntr.device), nte, yte.to(ntr.device), c["k"], c["T"], num_classes=c["classes"])
        if dev != "cpu":
            GU.record_parity(test="ref_checkpoint_" + key, prec=str(E.ops.act_dtype()), top=top, ref_top=gold[key]["top"],
                             feat_rel=max(((f.cpu() - w).abs().max() / w.abs().max()).item() for f, w in zip(feats, (gold[key]["train"], gold[key]["test"]))))
        for f, want in zip(feats, (gold[key]["train"], gold[key]["test"])):
            assert (f.cpu() - want).abs().max().item() <= tol * want.abs().max().item(), (key, (f.cpu() - want).abs().max().item())
        assert abs(top[0] - gold[key]["top"][0]) <= top_tol and abs(top[1] - gold[key]["top"][1]) <= top_tol, (key, top, gold[key]["top"])
    assert E.load_pretrained_weights(model, path + ".absent", "teacher") is None  # (no file: the random initialisation stays)


def test_reference_written_checkpoint_through_the_knn_consumers(cpu_ops, monkeypatch):
    from esvit_amd import eval as E
    monkeypatch.setattr(E, "ops", cpu_ops)
    check_ref_checkpoint()


@pytest.mark.parametrize("name", sorted(GU.FULL_CFG_CASES))
def test_baseline_configs_3_to_5_full_width_composition_matches_reference_golden(name, cpu_ops):
    """BASELINE configs 3-5 at FULL width through the product's host code (kernels replaced by their torch restatement, fp32) vs
    the fixtures produced by the reference's own modules from its own experiment yamls (tests/golden/full_configs.pt): module
    tree / state_dict layout, outputs, loss, centres, every gradient norm, sampled gradient tensors"""
    from tests.test_step_gpu import FULL_CFG_GOLD, full_case_deltas, run_full_cfg_case
    g = torch.load(FULL_CFG_GOLD, map_location="cpu", weights_only=False)[name]
    student, loss_fn, s_out, t_out, loss = run_full_cfg_case(name, torch.device("cpu"))
    assert [k for k, _ in student.named_parameters()] == g["param_names"]
    assert [(k, tuple(v.shape)) for k, v in student.state_dict().items()] == g["keys"]
    assert list(s_out[3]) == g["npatch"]
    out_rel, norm_rel, worst, worst_name = full_case_deltas(g, student, s_out, GU.FULL_CFG_SAMPLE)
    assert out_rel < 1e-4, out_rel
    assert abs(loss.item() - g["loss"]) < 1e-4, (loss.item(), g["loss"])
    assert norm_rel < 2e-3 and worst < 5e-3, (norm_rel, worst_name, worst)
    assert (loss_fn.center - g["center"]).abs().max().item() < 1e-6 and (loss_fn.center_grid - g["center_grid"]).abs().max().item() < 1e-6


def test_fused_mlp_branch_host_logic_matches_reference_golden(cpu_ops, monkeypatch):
    """the training path of the fused MLP branch (esvit_mlp_fused_fwd / _bwd, LayerNorm folded out of the fc1 weight gradient by
    esvit_ln_fold_finish) through the product's autograd glue, kernels replaced by their fp32 restatement: Swin-T at full width
    (stages 0 / 1 take the fused branch) reproduces the reference's own step -- loss, every gradient norm, sampled gradients"""
    import esvit_amd.functional as Fn
    from tests.test_step_gpu import FULL_GOLD, full_case_deltas, run_full_case
    monkeypatch.setattr(ops_ref, "mlp_fused_supported", lambda dt, C, backward=False: C in (96, 192))
    calls = {"fwd": 0, "bwd": 0}
    f0, b0 = ops_ref.mlp_fused_fwd, ops_ref.mlp_fused_bwd
    monkeypatch.setattr(ops_ref, "mlp_fused_fwd", lambda *a, **k: (calls.__setitem__("fwd", calls["fwd"] + 1), f0(*a, **k))[1])
    monkeypatch.setattr(ops_ref, "mlp_fused_bwd", lambda *a, **k: (calls.__setitem__("bwd", calls["bwd"] + 1), b0(*a, **k))[1])
    assert Fn.MLP_FUSED_TRAIN
    g = torch.load(FULL_GOLD, map_location="cpu", weights_only=False)["swin_t_k8192_b2"]
    student, loss_fn, s_out, t_out, loss = run_full_case("swin_t_k8192_b2", torch.device("cpu"))
    assert calls["bwd"] == 4 and calls["fwd"] == 8, calls  # stages 0 and 1, two blocks each: student (fwd + bwd) and teacher (fwd)
    out_rel, norm_rel, worst, worst_name = full_case_deltas(g, student, s_out)
    assert out_rel < 1e-4 and abs(loss.item() - g["loss"]) < 1e-4, (out_rel, loss.item(), g["loss"])
    assert norm_rel < 2e-3 and worst < 5e-3, (norm_rel, worst_name, worst)


def test_fused_attention_branch_host_logic_matches_reference_golden(cpu_ops, monkeypatch):
    """the fused attention branch (esvit_attn_branch_fwd) through the product's autograd glue -- per-group row slices of the stage's
    matrices, side outputs for the unfused backward, DropPath row scales, the LayerNorm hand-over it replaces -- with the kernel
    replaced by its fp32 restatement: Swin-T at full width reproduces the reference's own step.  The teacher takes the kernel in
    stages 0 and 1 (no side outputs), the student in stage 0 (C = 96) with side outputs"""
    import esvit_amd.functional as Fn
    from tests.test_step_gpu import FULL_GOLD, full_case_deltas, run_full_case
    monkeypatch.setattr(ops_ref, "attn_branch_supported", lambda dt, C, nH, N, rows=0, windows=0: C in (96, 192) and C == 32 * nH and N <= 64)
    calls = {"plain": 0, "save": 0}
    f0 = ops_ref.attn_branch_fwd

    def counted(*a, **k):
        calls["save" if k.get("save") else "plain"] += 1
        return f0(*a, **k)
    monkeypatch.setattr(ops_ref, "attn_branch_fwd", counted)
    assert Fn.ATTN_FUSED
    g = torch.load(FULL_GOLD, map_location="cpu", weights_only=False)["swin_t_k8192_b2"]
    student, loss_fn, s_out, t_out, loss = run_full_case("swin_t_k8192_b2", torch.device("cpu"))
    # student: stage 0, two blocks x two resolution groups; teacher: stages 0 and 1, two blocks each, one group
    assert calls == {"plain": 4, "save": 4}, calls
    out_rel, norm_rel, worst, worst_name = full_case_deltas(g, student, s_out)
    assert out_rel < 1e-4 and abs(loss.item() - g["loss"]) < 1e-4, (out_rel, loss.item(), g["loss"])
    assert norm_rel < 2e-3 and worst < 5e-3, (norm_rel, worst_name, worst)


@pytest.mark.parametrize("name", sorted(GU.FULL_VIL_CASES))
def test_vil_full_width_composition_matches_reference_golden(name, cpu_ops):
    """Vision Longformer (vil_tiny) through the product's host code -- module tree, patch embeddings with resampled position
    embeddings, the chunk-neighbourhood attention, block pairs -- with the kernels replaced by their fp32 restatement, vs the
    fixture produced by the reference's own MsViT (sliding-chunk implementation of layers/) from its own yaml"""
    from tests.test_step_gpu import check_full_vil_case
    check_full_vil_case(name, torch.device("cpu"), True, (1e-4, 1e-4, 2e-3, 5e-3))


def test_vil_ragged_route_equals_reference_schedule(cpu_ops):
    """the full-attention stages over the rows of all resolution groups at once (MsViT._forward_ragged) vs one pass per group
    (vision_longformer.py:699-752): same outputs and gradients"""
    import esvit_amd
    from esvit_amd import config as CFG
    arch = 'l1,h1,d32,n1,s1,g1,p4,f7_l2,h2,d64,n1,s1,g1,p2,f7_l3,h2,d64,n2,s0,g1,p2,f7_l4,h2,d64,n1,s0,g0,p2,f7'
    cfg = CFG.vil_config("vil_tiny", arch=arch, DROP_PATH=0.0)
    crops = [torch.randn(2, 3, 112, 112) for _ in range(2)] + [torch.randn(2, 3, 48, 48) for _ in range(3)]
    outs, grads = [], []
    for ragged in (True, False):
        torch.manual_seed(0)
        m = esvit_amd.build_model(cfg, use_dense_prediction=True)
        GU.fill_state_dict(m.state_dict(), 3)
        m.head_dense = torch.nn.Identity()
        m.ragged_multi_crop = ragged
        o = m([c.clone() for c in crops])
        (o[0].square().sum() + o[1].square().mean()).backward()
        outs.append(o)
        grads.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    assert outs[0][3] == outs[1][3]
    for a, b in zip(outs[0][:3], outs[1][:3]):
        assert (a - b).abs().max().item() < 1e-5 * (1 + b.abs().max().item())
    assert set(grads[0]) == set(grads[1])
    for k in grads[0]:
        assert (grads[0][k] - grads[1][k]).norm().item() < 1e-4 * (1e-6 + grads[1][k].norm().item()), k


def test_vil_refuses_what_is_not_built():
    from esvit_amd import config as CFG
    import esvit_amd
    with pytest.raises(NotImplementedError):
        esvit_amd.build_model(CFG.vil_config("vil_tiny", arch='l1,h1,d48,n1,s1,g1,p4,f7,a0_l2,h3,d96,n1,s1,g1,p2,f7_l3,h3,d192,n1,s0,g1,p2,f7'))
    cfg = CFG.vil_config("vil_tiny")
    cfg["MODEL"]["SPEC"]["MSVIT"]["SHARE_W"] = False
    with pytest.raises(NotImplementedError):
        esvit_amd.build_model(cfg)
    assert esvit_amd.models.is_model("cls_vil") and esvit_amd.models.is_model("vision_longformer")


def test_logit_statistics_handover_host_logic_matches_reference_golden(cpu_ops, monkeypatch):
    """the softmax row statistics the heads' last-layer GEMM emits for the loss (esvit_gemm_desc::rowstat -> loss.arm_logit_stats,
    DINOHead.logit_stats, the `esvit_row_stats` attribute, the token check): with the kernel replaced by its restatement and the shape
    gate opened, all four logit tensors arrive with statistics, the loss uses them, and the step still reproduces the reference's"""
    import esvit_amd.loss as L
    from tests.test_step_gpu import FULL_GOLD, full_case_deltas, run_full_case
    monkeypatch.setattr(ops_ref, "row_stats_supported", lambda dt, M, N: True)
    taken = []
    f0 = L._taken_stats
    monkeypatch.setattr(L, "_taken_stats", lambda x, tok: (taken.append(f0(x, tok) is not None), f0(x, tok))[1])
    g = torch.load(FULL_GOLD, map_location="cpu", weights_only=False)["swin_t_k8192_b2"]
    student, loss_fn, s_out, t_out, loss = run_full_case("swin_t_k8192_b2", torch.device("cpu"), armed=True)
    assert taken == [True] * 4, taken
    assert student.head.logit_stats is None and student.head_dense.logit_stats is None  # disarmed: later forwards emit nothing
    out_rel, norm_rel, worst, worst_name = full_case_deltas(g, student, s_out)
    assert out_rel < 1e-4 and abs(loss.item() - g["loss"]) < 1e-4, (out_rel, loss.item(), g["loss"])
    assert norm_rel < 2e-3 and worst < 5e-3, (norm_rel, worst_name, worst)
    # statistics computed for other centre values are refused: a centre update between the forwards and the loss changes the token
    taken.clear()
    loss_fn.arm_logit_stats(student, student, 0)
    crops = [torch.randn(2, 3, 224, 224), torch.randn(2, 3, 224, 224)]
    with torch.no_grad():
        t2 = student(crops)
    loss_fn.disarm_logit_stats(student, student)
    loss_fn.update_center(t2[0], t2[1])
    assert L._taken_stats(t2[0], loss_fn._token("t0", 1.0 / float(loss_fn.teacher_temp_schedule[0]))) is None


def check_linear_probe(dev="cpu", tol=2e-4):
    """eval_linear.py's probe (train + validate_network + LinearClassifier) through esvit_amd.eval over our backbone's
    forward_return_n_last_blocks vs the fixture produced by the reference's own functions on its own backbone"""
    from esvit_amd import eval as E
    g = torch.load(os.path.join(GOLD, "linear_probe.pt"), weights_only=False)
    c = GU.LINEAR_PROBE
    model = build_nano()
    GU.fill_state_dict(model.state_dict(), 0)
    model = model.to(dev).eval()
    clf = E.LinearClassifier(g["dim"], c["num_labels"])
    assert list(clf.state_dict().keys()) == g["keys"]
    GU.linear_probe_init(clf)
    clf = clf.to(dev)
    opt = torch.optim.SGD(clf.parameters(), c["lr"], momentum=0.9, weight_decay=0)
    tr, va = GU.linear_probe_data()
    depths = list(GU.NANO["depths"])
    stats = [E.train_linear_epoch(model, clf, opt, tr, ep, c["n_last_blocks"], c["avgpool"], depths) for ep in range(2)]
    val = E.validate_network(va, model, clf, c["n_last_blocks"], c["avgpool"], depths)
    for got, want in zip(stats, g["train"]):
        assert abs(got["loss"] - want["loss"]) < tol * 10 and abs(got["lr"] - want["lr"]) < 1e-9, (got, want)
    assert abs(val["loss"] - g["val"]["loss"]) < tol * 10 and val["acc1"] == pytest.approx(g["val"]["acc1"], abs=1e-3), (val, g["val"])
    assert val["acc5"] == pytest.approx(g["val"]["acc5"], abs=1e-3)
    assert (clf.linear.weight.detach().cpu() - g["weight"]).abs().max().item() < tol
    assert (clf.linear.bias.detach().cpu() - g["bias"]).abs().max().item() < tol
    # a probe checkpoint written by the reference's loop (eval_linear.py:232-238: "state_dict") loads
    clf2 = E.LinearClassifier(g["dim"], c["num_labels"])
    clf2.load_state_dict({"linear.weight": g["weight"], "linear.bias": g["bias"]})


def test_linear_probe_host_logic_matches_reference_golden(cpu_ops):
    check_linear_probe("cpu")
