"""SURVEY.md 8f-3 step variants against fixtures produced by the reference's own code (tests/golden/variants.pt,
oracle/gen_golden.py:gen_variants): DINOHead(use_bn=True) -- the oracle restatement and the product's host logic on CPU,
the HIP path on the GPU."""
import os

import pytest
import torch

from oracle import esvit_oracle as O
from oracle import ops_ref
from tests import golden_utils as GU

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "variants.pt")


@pytest.fixture(scope="module")
def gold():
    return torch.load(GOLD, map_location="cpu", weights_only=False)


def _bn_head(device=None):
    import esvit_amd
    c = GU.BN_HEAD
    head = esvit_amd.DINOHead(c["in_dim"], c["out_dim"], use_bn=True, hidden_dim=c["hidden_dim"], bottleneck_dim=c["bottleneck_dim"])
    GU.fill_bn_head(head.state_dict(), 31)
    head.sync_bn_group = False
    return head.to(device) if device is not None else head


def _check_bn_head(head, g, dev, tol):
    x, probe = GU.bn_head_inputs()
    x = x.to(dev).requires_grad_(True)
    head.train()
    out = head(x)
    (out * probe.to(dev)).sum().backward()

    def rel(a, b):  # (the Linear biases in front of a BatchNorm have an exactly-zero gradient: floor the denominator)
        return ((a.detach().float().cpu() - b).norm() / (b.norm() + 1.0)).item()

    assert rel(out, g["logits"]) < tol, rel(out, g["logits"])
    assert rel(x.grad, g["dx"]) < 4 * tol, rel(x.grad, g["dx"])
    for n, p in head.named_parameters():
        if n in ("mlp.0.bias", "mlp.3.bias"):
            # a bias in front of a BatchNorm has an exactly-zero gradient; what arrives is rounding noise of the column sum
            wn = dict(head.named_parameters())[n.replace("bias", "weight")].grad.norm().item()
            assert p.grad.norm().item() < max(20 * tol * wn, 1e-4), (n, p.grad.norm().item(), wn)
        elif n in g["grads"]:
            assert rel(p.grad, g["grads"][n]) < 4 * tol, (n, rel(p.grad, g["grads"][n]))
        else:
            assert p.grad is None, n
    for n, b in head.named_buffers():
        want = g["buffers_after"][n]
        if b.is_floating_point():
            assert rel(b, want) < tol, (n, rel(b, want))
        else:
            assert int(b) == int(want), n  # num_batches_tracked
    head.eval()
    with torch.no_grad():
        ev = head(x.detach())
    assert rel(ev, g["logits_eval"]) < tol, rel(ev, g["logits_eval"])


def test_oracle_bn_head_matches_reference_golden(gold):
    g = gold["bn_head"]
    c = GU.BN_HEAD
    import esvit_amd  # the module only supplies the state_dict layout (reference names)
    sd = {k: v.clone() for k, v in _bn_head().state_dict().items()}
    assert set(k for k in sd if "num_batches" not in k) == set(k for k in list(g["grads"]) + list(g["buffers_after"]) + ["last_layer.weight_g"]
                                                              if "num_batches" not in k)
    x, probe = GU.bn_head_inputs()
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k in g["grads"]}
    full = dict(sd)
    full.update(leaf)
    xr = x.clone().requires_grad_(True)
    out = O.dino_head_bn(full, "", xr, training=True)
    (out * probe).sum().backward()
    assert (out - g["logits"]).abs().max().item() < 1e-5
    assert (xr.grad - g["dx"]).abs().max().item() < 1e-5
    for n, t in leaf.items():
        assert (t.grad - g["grads"][n]).abs().max().item() < 2e-5 * (1 + g["grads"][n].abs().max().item()), n
    for n in ("mlp.1.running_mean", "mlp.1.running_var", "mlp.4.running_mean", "mlp.4.running_var"):
        assert (full[n] - g["buffers_after"][n]).abs().max().item() < 1e-6, n
    with torch.no_grad():
        ev = O.dino_head_bn(full, "", x, training=False)
    assert (ev - g["logits_eval"]).abs().max().item() < 1e-5


def test_bn_head_host_logic_cpu(gold, monkeypatch, lib_built):
    """the product's DINOHead(use_bn=True) module and autograd function with every op swapped for its torch restatement"""
    import esvit_amd
    import esvit_amd.functional as Fn
    import esvit_amd.params as P
    esvit_amd.set_precision("fp32")
    for mod in (Fn, P):
        monkeypatch.setattr(mod, "ops", ops_ref)
    _check_bn_head(_bn_head(), gold["bn_head"], torch.device("cpu"), 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_bn_head_gpu(gold, prec, lib_built):
    import esvit_amd
    esvit_amd.set_precision(prec)
    try:
        dev = torch.device("cuda:0")
        _check_bn_head(_bn_head(dev), gold["bn_head"], dev, 2e-5 if prec == "fp32" else 2e-2)
    finally:
        esvit_amd.set_precision("bf16")


# ---- LARS / SGD (main_esvit.py:412-415, utils.py:519-557) ---------------------------------------------------------------
def _oracle_lars_run():
    p, grads = GU.lars_case()
    p = {n: t.clone() for n, t in p.items()}
    mu = {n: torch.zeros_like(t) for n, t in p.items()}
    for (lr, wd), gs in zip(GU.LARS_SCHED, grads):
        for n in p:
            p[n], mu[n] = O.lars_update(p[n], O.clip_gradient(gs[n], 3.0), mu[n], lr, wd if p[n].ndim != 1 else 0.0)
    return p, mu


def test_oracle_lars_matches_reference_golden(gold):
    p, mu = _oracle_lars_run()
    for n in p:
        assert (p[n] - gold["lars"]["params"][n]).abs().max().item() < 1e-6, n
        assert (mu[n] - gold["lars"]["mu"][n]).abs().max().item() < 1e-6, n


class _Holder(torch.nn.Module):
    def __init__(self, tensors):
        super().__init__()
        for n, t in tensors.items():
            self.register_parameter(n, torch.nn.Parameter(t.clone()))


@pytest.mark.gpu
def test_fused_lars_matches_reference_golden_gpu(gold, lib_built):
    """the fused clip -> LARS -> EMA update on the fixture of the reference's utils.LARS (+ utils.clip_gradients)"""
    from esvit_amd.update import FusedClipAdamWEMA
    dev = torch.device("cuda:0")
    p0, grads = GU.lars_case()
    student, teacher = _Holder(p0).to(dev), _Holder({n: t * 0.5 for n, t in p0.items()}).to(dev)
    upd = FusedClipAdamWEMA(student, teacher, rule="lars")
    t_ref = {n: t * 0.5 for n, t in p0.items()}
    ref_p = {n: t.clone() for n, t in p0.items()}
    ref_mu = {n: torch.zeros_like(t) for n, t in p0.items()}
    for (lr, wd), gs in zip(GU.LARS_SCHED, grads):
        for n, prm in student.named_parameters():
            prm.grad = gs[n].to(dev)
        upd.step(lr, wd, 0.9, clip_grad=3.0)
        for n in ref_p:
            ref_p[n], ref_mu[n] = O.lars_update(ref_p[n], O.clip_gradient(gs[n], 3.0), ref_mu[n], lr, wd if ref_p[n].ndim != 1 else 0.0)
            t_ref[n] = t_ref[n] * 0.9 + ref_p[n] * 0.1
    sd = upd.state_dict()
    for n, prm in student.named_parameters():
        assert (prm.detach().cpu() - gold["lars"]["params"][n]).abs().max().item() < 2e-6, n
    for n, prm in teacher.named_parameters():
        assert (prm.detach().cpu() - t_ref[n]).abs().max().item() < 2e-6, n
    order = [n for n in p0 if p0[n].ndim != 1] + [n for n in p0 if p0[n].ndim == 1]  # get_params_groups: regularised first
    for k, n in enumerate(order):
        assert set(sd["state"][k]) == {"mu"}
        assert (sd["state"][k]["mu"].cpu() - gold["lars"]["mu"][n]).abs().max().item() < 2e-6, n


@pytest.mark.gpu
def test_fused_sgd_matches_torch_gpu(lib_built):
    """clip + torch.optim.SGD(lr=0, momentum=0.9) (main_esvit.py:413) + EMA, three steps, through the binding the drop-in uses"""
    import copy
    from esvit_amd.update import bind_torch_optimizer, get_params_groups
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(300, 77), torch.nn.LayerNorm(77), torch.nn.Linear(77, 5000)).to(dev)
    s_ref, t_ref, s_fus, t_fus = (copy.deepcopy(net) for _ in range(4))
    for t in (t_ref, t_fus):
        for p in t.parameters():
            p.data.mul_(0.5)
    opt_ref = torch.optim.SGD(get_params_groups(s_ref), lr=0, momentum=0.9)
    opt_fus = torch.optim.SGD(get_params_groups(s_fus), lr=0, momentum=0.9)
    upd = bind_torch_optimizer(opt_fus, s_fus, t_fus)
    assert upd.rule == "sgd"
    for it in range(3):
        lr, wd, m = 0.05 * (it + 1), 0.04 + 0.01 * it, 0.99
        grads = [torch.randn_like(p) * (10.0 if i == 0 else 0.01) for i, p in enumerate(s_ref.parameters())]
        for (p1, p2, g) in zip(s_ref.parameters(), s_fus.parameters(), grads):
            p1.grad, p2.grad = g.clone(), g.clone()
        for i, pg in enumerate(opt_ref.param_groups):
            pg["lr"] = lr
            if i == 0:
                pg["weight_decay"] = wd
        for p in s_ref.parameters():
            coef = 3.0 / (p.grad.norm(2) + 1e-6)
            if coef < 1:
                p.grad.mul_(coef)
        opt_ref.step()
        with torch.no_grad():
            for q, k in zip(s_ref.parameters(), t_ref.parameters()):
                k.mul_(m).add_((1 - m) * q.detach())
        upd.step(lr, wd, m, clip_grad=3.0)
    for a, b in zip(s_ref.parameters(), s_fus.parameters()):
        assert (a - b).abs().max().item() < 2e-5 * (1 + a.abs().max().item())
    for a, b in zip(t_ref.parameters(), t_fus.parameters()):
        assert (a - b).abs().max().item() < 2e-5 * (1 + a.abs().max().item())
    # the caller's optimizer holds the live momentum buffers and schedule values (its checkpoint is the real state)
    assert opt_fus.param_groups[0]["lr"] == lr and opt_fus.param_groups[0]["weight_decay"] == wd
    for pr, pf in zip(s_ref.parameters(), s_fus.parameters()):
        br, bf = opt_ref.state[pr]["momentum_buffer"], opt_fus.state[pf]["momentum_buffer"]
        assert (br - bf).abs().max().item() < 2e-5 * (1 + br.abs().max().item())


# ---- DINOLoss with mixup targets (main_esvit.py:518-534, 639-641) -----------------------------------------------------------
def test_oracle_mixup_loss_matches_reference_golden(gold):
    s, t, c0, T = GU.mixup_case()
    s = s.clone().requires_grad_(True)
    loss, _ = O.dino_loss(s, t, c0, O.teacher_temp(2, 0.04, 0.07, 5, 10), GU.MIXUP["ncrops"], targets_mixup=T)
    loss.backward()
    assert abs(loss.item() - gold["mixup"]["loss"]) < 1e-6
    assert (s.grad - gold["mixup"]["ds"]).abs().max().item() < 1e-7


def _check_mixup_loss(dev, gold, tol):
    import esvit_amd
    mc = GU.MIXUP
    s, t, c0, T = GU.mixup_case()
    lf = esvit_amd.DINOLoss(mc["K"], mc["ncrops"], 0.04, 0.07, 5, 10).to(dev)
    lf.center.copy_(c0.to(dev))
    s = s.to(dev).requires_grad_(True)
    loss = lf(s, t.to(dev), 2, [m.to(dev) for m in T])
    (loss * 3.0).backward()  # a non-unit grad_output exercises the rescale pass as well
    lf.synchronize()
    assert abs(loss.item() - gold["mixup"]["loss"]) < tol * 10
    assert (s.grad.float().cpu() / 3.0 - gold["mixup"]["ds"]).abs().max().item() < tol
    assert (lf.center.cpu() - gold["mixup"]["center_after"]).abs().max().item() < 1e-5
    # without targets the same module still gives the plain loss (the static two-term tables are not disturbed)
    plain, _ = O.dino_loss(gold["_s"], gold["_t"], gold["mixup"]["center_after"], O.teacher_temp(2, 0.04, 0.07, 5, 10), mc["ncrops"])
    got = lf(gold["_s"].to(dev), gold["_t"].to(dev), 2, None)
    assert abs(got.item() - plain.item()) < tol * 10


def test_mixup_loss_host_logic_cpu(gold, monkeypatch, lib_built):
    import esvit_amd
    import esvit_amd.loss as L
    esvit_amd.set_precision("fp32")
    monkeypatch.setattr(L, "ops", ops_ref)
    g = dict(gold)
    g["_s"], g["_t"] = GU.mixup_case()[:2]
    _check_mixup_loss(torch.device("cpu"), g, 5e-6)


@pytest.mark.gpu
def test_mixup_loss_gpu(gold, lib_built):
    import esvit_amd
    esvit_amd.set_precision("fp32")
    try:
        g = dict(gold)
        g["_s"], g["_t"] = GU.mixup_case()[:2]
        _check_mixup_loss(torch.device("cuda:0"), g, 5e-6)
    finally:
        esvit_amd.set_precision("bf16")


# ---- label-smoothed mixup targets (--smoothing > 0, main_esvit.py:230): dense [B, B] matrices = sparse part + one constant per crop
def _smoothing_gold():
    return torch.load(os.path.join(os.path.dirname(GOLD), "mixup_smoothing.pt"), weights_only=False)


def test_oracle_smoothed_mixup_loss_matches_reference_golden():
    g = _smoothing_gold()
    s, t, c0, T = GU.mixup_case(GU.MIXUP_SMOOTHING)
    s = s.clone().requires_grad_(True)
    loss, _ = O.dino_loss(s, t, c0, O.teacher_temp(2, 0.04, 0.07, 5, 10), GU.MIXUP["ncrops"], targets_mixup=T)
    loss.backward()
    assert abs(loss.item() - g["loss"]) < 2e-6 and (s.grad - g["ds"]).abs().max().item() < 1e-7


def _check_smoothed_mixup_loss(dev, tol):
    import esvit_amd
    g, mc = _smoothing_gold(), GU.MIXUP
    s, t, c0, T = GU.mixup_case(GU.MIXUP_SMOOTHING)
    lf = esvit_amd.DINOLoss(mc["K"], mc["ncrops"], 0.04, 0.07, 5, 10).to(dev)
    lf.center.copy_(c0.to(dev))
    s = s.to(dev).requires_grad_(True)
    loss = lf(s, t.to(dev), 2, [m.to(dev) for m in T])
    loss.backward()
    lf.synchronize()
    assert abs(loss.item() - g["loss"]) < tol * 10 * max(1.0, abs(g["loss"])), (loss.item(), g["loss"])
    assert (s.grad.float().cpu() - g["ds"]).abs().max().item() < tol * max(1.0, g["ds"].abs().max().item())
    assert (lf.center.cpu() - g["center_after"]).abs().max().item() < 1e-5
    # a matrix that is neither sparse nor sparse + constant is refused (on a fresh module: the structure is checked once per run)
    lf2 = esvit_amd.DINOLoss(mc["K"], mc["ncrops"], 0.04, 0.07, 5, 10).to(dev)
    bad = [torch.rand(mc["B"], mc["B"], device=dev) for _ in T]
    with pytest.raises(NotImplementedError):
        lf2(s.detach(), t.to(dev), 2, bad)


def test_smoothed_mixup_loss_host_logic_cpu(monkeypatch, lib_built):
    import esvit_amd
    import esvit_amd.loss as L
    esvit_amd.set_precision("fp32")
    monkeypatch.setattr(L, "ops", ops_ref)
    _check_smoothed_mixup_loss(torch.device("cpu"), 5e-6)


@pytest.mark.gpu
def test_smoothed_mixup_loss_gpu(lib_built):
    import esvit_amd
    esvit_amd.set_precision("fp32")
    try:
        _check_smoothed_mixup_loss(torch.device("cuda:0"), 5e-6)
    finally:
        esvit_amd.set_precision("bf16")


@pytest.mark.gpu
def test_train_one_epoch_mixup_drop_in_gpu(lib_built):
    """engine.train_one_epoch with a mixup_fn (main_esvit.py:515-544): the first num_mixup_views crops are mixed, the teacher
    sees the un-mixed global views, the target matrices reach DINOLoss.  Checked against the oracle on the logits the HIP
    path produced, and against an explicit EsvitTrainer.step with the same mixed inputs."""
    import argparse
    import esvit_amd
    from esvit_amd import engine
    from esvit_amd.update import get_params_groups
    from tests.test_composition_cpu import build_nano_view
    esvit_amd.set_precision("fp32")
    try:
        dev = torch.device("cuda:0")
        K, B = GU.NANO_HEAD["out_dim"], 4

        def fresh():
            s, t = build_nano_view(), build_nano_view(teacher=True)
            GU.fill_state_dict(s.state_dict(), 0)
            GU.fill_state_dict(t.state_dict(), 7)
            s.head.last_layer.weight_g.data.fill_(1)
            for p in t.parameters():
                p.requires_grad = False
            return s.to(dev), t.to(dev), esvit_amd.DINOLoss(K, 10, 0.04, 0.07, 5, 10).to(dev)

        lams = iter([0.8, 0.3, 0.8, 0.3])

        def mixup_fn(x, targets):  # timm.data.Mixup in 'batch' mode with a fixed lambda sequence
            lam = next(lams)
            onehot = torch.nn.functional.one_hot(targets, B).float()
            return x * lam + x.flip(0) * (1 - lam), onehot * lam + onehot.flip(0) * (1 - lam)

        crops = [c.to(dev) for c in GU.make_crops(B, seed=5)]
        args = argparse.Namespace(clip_grad=3.0, freeze_last_layer=0, batch_size_per_gpu=B, num_mixup_views=2)

        class Loader:
            sampler = None

            def __len__(self):
                return 1

            def __iter__(self):
                return iter([(crops, None)])

        s1, t1, l1 = fresh()
        o1 = torch.optim.AdamW(get_params_groups(s1))
        init = {n: p.detach().clone() for n, p in s1.named_parameters()}
        st = engine.train_one_epoch(s1, t1, t1, l1, Loader(), o1, [1e-3], [0.04], [0.99], 0, mixup_fn, None, args)
        # explicit: same mixing, oracle loss on the HIP logits
        s2, t2, l2 = fresh()
        mixed, T = [], []
        for i, c in enumerate(crops):
            if i < 2:
                m, tt = mixup_fn(c, torch.arange(B, device=dev))
            else:
                m, tt = c, torch.eye(B, device=dev)
            mixed.append(m)
            T.append(tt)
        with torch.no_grad():
            s_log, t_log = s2(mixed), t2(crops[:2])
            want, _ = O.dino_loss(s_log.cpu(), t_log.cpu(), torch.zeros(1, K), O.teacher_temp(0, 0.04, 0.07, 5, 10), 10,
                                  targets_mixup=[m.cpu() for m in T])
        assert abs(st["loss"] - want.item()) < 2e-4, (st["loss"], want.item())
        tr = engine.EsvitTrainer(s2, t2, l2, clip_grad=3.0, freeze_last_layer=0)
        tr.step(mixed, 1e-3, 0.04, 0.99, 0, teacher_images=crops[:2], targets_mixup=T)
        for (n, a), (_, b) in zip(s1.named_parameters(), s2.named_parameters()):
            ua, ub = (a - init[n]).detach(), (b - init[n]).detach()
            assert (ua - ub).norm().item() <= 2e-2 * ub.norm().item() + 1e-9, n
    finally:
        esvit_amd.set_precision("bf16")


# ---- USE_APE (swin_transformer.py:623-627, 680-681) -----------------------------------------------------------------------
def _ape_model():
    from esvit_amd import models
    from oracle import ref_loader as RL
    cfg = RL.swin_config(embed_dim=GU.NANO["embed_dim"], depths=(2, 2, 2), heads=(1, 2, 4), window=GU.NANO["window"], img=112)
    cfg.MODEL["SPEC"]["USE_APE"] = True
    m = models.build_model(cfg, is_teacher=False, use_dense_prediction=False)
    GU.fill_state_dict(m.state_dict(), 17)
    return m


def _check_ape(m, g, dev, tol):
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == g["keys"]  # absolute_pos_embed sits where the reference puts it
    x, pr = GU.ape_inputs(m.num_features)
    cls = m.forward_features(x.to(dev))
    (cls * pr.to(dev)).sum().backward()

    def rel(a, b):
        return ((a.detach().float().cpu() - b).norm() / (b.norm() + 1e-12)).item()
    assert rel(cls, g["cls"]) < tol, rel(cls, g["cls"])
    prm = dict(m.named_parameters())
    for n, want in g["grads"].items():
        assert rel(prm[n].grad, want) < 5 * tol, (n, rel(prm[n].grad, want))
    with pytest.raises(ValueError):  # another resolution: the reference's broadcast add fails too
        m.forward_features(torch.zeros(1, 3, 56, 56, device=dev))


def test_ape_host_logic_cpu(gold, monkeypatch, lib_built):
    import esvit_amd
    import esvit_amd.functional as Fn
    import esvit_amd.params as P
    esvit_amd.set_precision("fp32")
    for mod in (Fn, P):
        monkeypatch.setattr(mod, "ops", ops_ref)
    _check_ape(_ape_model(), gold["ape"], torch.device("cpu"), 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_ape_gpu(gold, prec, lib_built):
    import esvit_amd
    esvit_amd.set_precision(prec)
    try:
        dev = torch.device("cuda:0")
        _check_ape(_ape_model().to(dev), gold["ape"], dev, 2e-5 if prec == "fp32" else 3e-2)
    finally:
        esvit_amd.set_precision("bf16")


# ---- PATCH_NORM False (swin_transformer.py:532-535, 545-546) ---------------------------------------------------------------
def _check_patch_norm(dev, tol):
    from esvit_amd import models
    from oracle import ref_loader as RL
    gold = torch.load(os.path.join(os.path.dirname(GOLD), "patch_norm.pt"), weights_only=False)
    for mode, g in gold.items():
        cfg = RL.swin_config(embed_dim=GU.NANO["embed_dim"], depths=(2, 2, 2), heads=(1, 2, 4), window=GU.NANO["window"], img=112)
        cfg.MODEL["SPEC"]["PATCH_NORM"] = False
        m = models.build_model(cfg, is_teacher=False, use_dense_prediction=False)
        assert m.patch_embed.norm is None
        assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == g["keys"]
        GU.fill_state_dict(m.state_dict(), 29)
        m = m.to(dev)
        xa, xb, pa, pab = (t.to(dev) for t in GU.patch_norm_inputs(m.num_features))
        if mode == "features":
            y = m.forward_features(xa)
            (y * pa).sum().backward()
        else:  # both resolutions through the ragged route: one embedding GEMM over the rows of both crops
            y = m([xa, xb])
            (y * pab).sum().backward()

        def rel(a, b):
            return ((a.detach().float().cpu() - b).norm() / (b.norm() + 1e-12)).item()
        assert rel(y, g["y"]) < tol, (mode, rel(y, g["y"]))
        prm = dict(m.named_parameters())
        for n, want in g["grads"].items():
            assert rel(prm[n].grad, want) < 5 * tol, (mode, n, rel(prm[n].grad, want))


def test_patch_norm_false_host_logic_cpu(monkeypatch, lib_built):
    import esvit_amd
    import esvit_amd.functional as Fn
    import esvit_amd.params as P
    esvit_amd.set_precision("fp32")
    for mod in (Fn, P):
        monkeypatch.setattr(mod, "ops", ops_ref)
    _check_patch_norm(torch.device("cpu"), 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_patch_norm_false_gpu(prec, lib_built):
    import esvit_amd
    esvit_amd.set_precision(prec)
    try:
        _check_patch_norm(torch.device("cuda:0"), 2e-5 if prec == "fp32" else 3e-2)
    finally:
        esvit_amd.set_precision("bf16")


def test_init_weights_resizes_bias_table_and_ape(tmp_path):
    """swin_transformer.py:852-917: a checkpoint of another window size / grid is loaded with bicubic resizing (closed form of
    :873-912; the reference's own init_weights cannot run this case, see the docstring of SwinTransformer.init_weights)"""
    from esvit_amd import models
    from oracle import ref_loader as RL

    def cfg(window, img):
        c = RL.swin_config(embed_dim=32, depths=(2, 2), heads=(1, 2), window=window, img=img)
        c.MODEL["SPEC"]["USE_APE"] = True
        return c
    src = models.build_model(cfg(7, 56), is_teacher=False, use_dense_prediction=False)
    GU.fill_state_dict(src.state_dict(), 3)
    ck = os.path.join(tmp_path, "ck.pth")
    torch.save(src.state_dict(), ck)
    dst = models.build_model(cfg(14, 112), is_teacher=False, use_dense_prediction=False)
    dst.init_weights(ck, ["*"], verbose=False)
    t7 = src.state_dict()["layers.0.blocks.0.attn.relative_position_bias_table"]
    t14 = dst.state_dict()["layers.0.blocks.0.attn.relative_position_bias_table"]
    want = torch.nn.functional.interpolate(t7.permute(1, 0).view(1, 1, 13, 13), size=(27, 27), mode="bicubic").view(1, 27 * 27).permute(1, 0)
    assert t14.shape == (27 * 27, 1) and torch.allclose(t14, want)
    a = dst.state_dict()["absolute_pos_embed"]
    assert a.shape == (1, 28 * 28, 32)
    assert torch.equal(dst.state_dict()["patch_embed.proj.weight"], src.state_dict()["patch_embed.proj.weight"])


# ---- PatchMerging on an odd feature map (swin_transformer.py:406-408) ------------------------------------------------------
def _check_oddmerge(g, dev, tol):
    from esvit_amd import models
    from oracle import ref_loader as RL
    cfg = RL.swin_config(embed_dim=GU.NANO["embed_dim"], depths=GU.NANO["depths"], heads=GU.NANO["heads"], window=GU.NANO["window"], img=112)
    m = models.build_model(cfg, is_teacher=False, use_dense_prediction=True)
    GU.fill_state_dict(m.state_dict(), 23)
    m = m.to(dev)
    x, pr = GU.ape_inputs(m.num_features)
    cls, region = m.forward_features(x.to(dev))
    assert region.shape == g["region"].shape  # 28 -> 14 -> 7 -> 4: sixteen tokens in the last stage
    ((cls * pr.to(dev)).sum() + region.sum() * 0.01).backward()

    def rel(a, b):
        return ((a.detach().float().cpu() - b).norm() / (b.norm() + 1e-12)).item()
    assert rel(cls, g["cls"]) < tol and rel(region, g["region"]) < tol, (rel(cls, g["cls"]), rel(region, g["region"]))
    prm = dict(m.named_parameters())
    for n, want in g["grads"].items():
        assert rel(prm[n].grad, want) < 5 * tol, (n, rel(prm[n].grad, want))


def test_odd_patch_merging_host_logic_cpu(gold, monkeypatch, lib_built):
    import esvit_amd
    import esvit_amd.functional as Fn
    import esvit_amd.params as P
    esvit_amd.set_precision("fp32")
    for mod in (Fn, P):
        monkeypatch.setattr(mod, "ops", ops_ref)
    _check_oddmerge(gold["oddmerge"], torch.device("cpu"), 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_odd_patch_merging_gpu(gold, prec, lib_built):
    import esvit_amd
    esvit_amd.set_precision(prec)
    try:
        _check_oddmerge(gold["oddmerge"], torch.device("cuda:0"), 2e-5 if prec == "fp32" else 3e-2)
    finally:
        esvit_amd.set_precision("bf16")


# ---- DINOHead(nlayers != 3) (vision_transformer.py:388-402) -------------------------------------------------------------
def _check_head_nlayers(dev, tol):
    import esvit_amd
    g = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "head_nlayers.pt"), weights_only=False)
    c = GU.HEAD_NLAYERS
    x, probe = GU.head_nlayers_inputs()
    for n in c["cases"]:
        head = esvit_amd.DINOHead(c["in_dim"], c["out_dim"], nlayers=n, hidden_dim=c["hidden_dim"], bottleneck_dim=c["bottleneck_dim"],
                                  norm_last_layer=False)
        assert [(k, tuple(v.shape)) for k, v in head.state_dict().items()] == g[n]["keys"], n
        GU.fill_state_dict(head.state_dict(), 60 + n)
        head = head.to(dev)
        xr = x.clone().to(dev).requires_grad_(True)
        y = head(xr)
        (y.float() * probe.to(dev)).sum().backward()
        rel = lambda a, b: ((a.float().cpu() - b).abs().max() / (b.abs().max() + 1e-12)).item()
        assert rel(y.detach(), g[n]["logits"]) < tol, (n, rel(y.detach(), g[n]["logits"]))
        assert rel(xr.grad, g[n]["dx"]) < 3 * tol, (n, rel(xr.grad, g[n]["dx"]))
        for k, p in head.named_parameters():
            assert rel(p.grad, g[n]["grads"][k]) < 3 * tol, (n, k, rel(p.grad, g[n]["grads"][k]))


def _check_head_bn_nlayers(dev, tol, zero_floor=1e-2):
    """DINOHead(use_bn=True) with 1, 2 and 4 layers (functional.DinoHeadBnNFn) vs the reference's own module in train mode"""
    import esvit_amd
    g = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "head_bn_nlayers.pt"), weights_only=False)
    c = GU.HEAD_NLAYERS
    x, probe = GU.head_nlayers_inputs()
    for n in GU.HEAD_BN_NLAYERS:
        head = esvit_amd.DINOHead(c["in_dim"], c["out_dim"], use_bn=True, nlayers=n, hidden_dim=c["hidden_dim"], bottleneck_dim=c["bottleneck_dim"],
                                  norm_last_layer=False)
        assert [(k, tuple(v.shape)) for k, v in head.state_dict().items()] == g[n]["keys"], n
        GU.fill_bn_head_n(head.state_dict(), 80 + n)
        head = head.to(dev).train()
        xr = x.clone().to(dev).requires_grad_(True)
        y = head(xr)
        (y.float() * probe.to(dev)).sum().backward()
        rel = lambda a, b: ((a.float().cpu() - b).abs().max() / (b.abs().max() + 1e-12)).item()
        assert rel(y.detach(), g[n]["logits"]) < tol, (n, rel(y.detach(), g[n]["logits"]))
        assert rel(xr.grad, g[n]["dx"]) < 3 * tol, (n, rel(xr.grad, g[n]["dx"]))
        gmax = max(v.abs().max().item() for v in g[n]["grads"].values())
        for k, p in head.named_parameters():
            # (the bias of a Linear in front of a BatchNorm has a mathematically zero gradient: rounding noise on both sides, so the
            # error is measured against the largest gradient of the head)
            err = (p.grad.float().cpu() - g[n]["grads"][k]).abs().max().item()
            assert err < 3 * tol * max(g[n]["grads"][k].abs().max().item(), zero_floor * gmax), (n, k, err)
        sd = head.state_dict()
        for k, want in g[n]["buffers"].items():
            assert torch.allclose(sd[k].float().cpu(), want.float(), rtol=max(tol, 1e-4), atol=max(tol, 1e-4)), (n, k)


def test_head_bn_nlayers_host_logic_cpu(monkeypatch, lib_built):
    import esvit_amd
    import esvit_amd.functional as Fn
    import esvit_amd.params as P
    esvit_amd.set_precision("fp32")
    for mod in (Fn, P):
        monkeypatch.setattr(mod, "ops", ops_ref)
    P.clear()
    _check_head_bn_nlayers(torch.device("cpu"), 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_head_bn_nlayers_gpu(prec, lib_built):
    import esvit_amd
    from esvit_amd import params as P
    esvit_amd.set_precision(prec)
    P.clear()
    try:
        # (bf16: the noise of the zero gradients is a sum of rounded rows, a percent of the head's largest gradient)
        _check_head_bn_nlayers(torch.device("cuda:0"), 2e-5 if prec == "fp32" else 2.5e-2, zero_floor=1e-2 if prec == "fp32" else 1.0)
    finally:
        esvit_amd.set_precision("bf16")


def test_head_nlayers_host_logic_cpu(monkeypatch, lib_built):
    """DINOHead with 1, 2 and 4 Linear layers (functional.DinoHeadNFn) vs the reference's own module (tests/golden/head_nlayers.pt)"""
    import esvit_amd
    import esvit_amd.functional as Fn
    import esvit_amd.params as P
    esvit_amd.set_precision("fp32")
    for mod in (Fn, P):
        monkeypatch.setattr(mod, "ops", ops_ref)
    P.clear()
    _check_head_nlayers(torch.device("cpu"), 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_head_nlayers_gpu(prec, lib_built):
    import esvit_amd
    from esvit_amd import params as P
    esvit_amd.set_precision(prec)
    P.clear()
    try:
        _check_head_nlayers(torch.device("cuda:0"), 2e-5 if prec == "fp32" else 2.5e-2)
    finally:
        esvit_amd.set_precision("bf16")
