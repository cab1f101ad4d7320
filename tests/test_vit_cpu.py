"""Monolithic ViT backbones (models/vision_transformer.py, the reference's default --arch), CPU side: module tree,
state_dict layout and the autograd glue (VitPatchEmbedFn / VitBlockFn / ApeAddFn / FinalNormFn) on the torch restatement of every
kernel, against tests/golden/nano_vit_step.pt -- produced by the REFERENCE's own VisionTransformer + DINOHead + DDINOLoss
(oracle/gen_golden.py vit)."""
import os
from functools import partial

import pytest
import torch

from oracle import ops_ref
from tests import golden_utils as GU
from tests.test_composition_cpu import cpu_ops, probe_close  # noqa: F401  (fixture)

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def build_nano_vit():
    from esvit_amd import models
    from esvit_amd.models import vision_transformer as V
    v = GU.NANO_VIT
    m = V.VisionTransformer(img_size=[v["sizes"][0]], patch_size=v["patch"], embed_dim=v["embed_dim"], depth=v["depth"], num_heads=v["heads"],
                            mlp_ratio=4, qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), drop_path_rate=0.0,
                            use_dense_prediction=True)
    hk = dict(hidden_dim=GU.NANO_HEAD["hidden_dim"], bottleneck_dim=GU.NANO_HEAD["bottleneck_dim"])
    m.head = models.DINOHead(v["embed_dim"], GU.NANO_HEAD["out_dim"], norm_last_layer=True, **hk)
    m.head_dense = models.DINOHead(v["embed_dim"], GU.NANO_HEAD["out_dim"], norm_last_layer=False, **hk)
    return m


def nano_vit_pair(dev="cpu"):
    student, teacher = build_nano_vit(), build_nano_vit()
    GU.fill_state_dict(student.state_dict(), 0)
    GU.fill_state_dict(teacher.state_dict(), 7)
    student.head.last_layer.weight_g.data.fill_(1)
    for p in teacher.parameters():
        p.requires_grad = False
    return student.to(dev), teacher.to(dev)


def run_nano_vit_step(student, teacher, loss_mod, dev="cpu"):
    crops = [c.to(dev) for c in GU.make_crops(2, n_local=3, sizes=GU.NANO_VIT["sizes"])]
    loss_fn = loss_mod.DDINOLoss(GU.NANO_HEAD["out_dim"], 5, 0.04, 0.07, 5, 10).to(dev)
    t_out = teacher(crops[:2])
    s_out = student(crops)
    loss = loss_fn(s_out, t_out, 2, None)
    loss.backward()
    return s_out, t_out, loss


def check_nano_vit(g, student, s_out, t_out, loss, rt, loss_tol, grad_tol):
    assert list(s_out[3]) == g["npatch"][0] and list(t_out[3]) == g["npatch"][1]
    for nm, t in (("s_cls", s_out[0]), ("s_reg", s_out[1]), ("s_fea", s_out[2]), ("t_cls", t_out[0]), ("t_reg", t_out[1]), ("t_fea", t_out[2])):
        probe_close(nm, t.float().cpu(), g[nm], rtol=rt)
    assert abs(loss.item() - g["ddino_loss"]) < loss_tol, (loss.item(), g["ddino_loss"])
    got = {n: p.grad for n, p in student.named_parameters() if p.grad is not None}
    assert sorted(got) == sorted(g["grad_norms"]) and [n for n, p in student.named_parameters() if p.grad is None] == g["no_grad"]
    for n, ref in g["grad_norms"].items():
        assert abs(got[n].norm().item() - ref) <= grad_tol * ref + 1e-9, (n, got[n].norm().item(), ref)
    for n, ref in g["grads"].items():
        probe_close("grad " + n, got[n].float().cpu(), ref, rtol=max(rt, grad_tol))


def check_nano_vit_hooks(g, student, dev="cpu", rt=3e-4):
    crops = [c.to(dev) for c in GU.make_crops(2, n_local=3, sizes=GU.NANO_VIT["sizes"])]
    with torch.no_grad():
        student.use_dense_prediction = False
        probe_close("view-level forward", student(crops).float().cpu(), g["view_only"], rtol=rt)
        student.use_dense_prediction = True
        feats = student.forward_return_n_last_blocks(crops[2], n=2, return_patch_avgpool=True)
        assert feats.shape == g["last_blocks"].shape and torch.allclose(feats.float().cpu(), g["last_blocks"], rtol=rt, atol=rt)
        attn = student.forward_selfattention(crops[0])
        assert attn.shape == g["last_attn"].shape and torch.allclose(attn.float().cpu(), g["last_attn"], rtol=rt, atol=1e-5)


def test_vit_state_dict_layout_matches_golden(lib_built):
    g = torch.load(os.path.join(GOLD, "nano_vit_step.pt"), weights_only=False)
    student = build_nano_vit()
    assert [(k, tuple(v.shape), str(v.dtype)) for k, v in student.state_dict().items()] == g["keys"]
    assert [n for n, _ in student.named_parameters()] == g["param_names"]
    from esvit_amd.models import vision_transformer as V
    for fn, (dim, heads, params) in {"deit_tiny": (192, 3, 5524416), "deit_small": (384, 6, 21665664), "vit_base": (768, 12, 85798656)}.items():
        m = getattr(V, fn)(patch_size=16, drop_path_rate=0.1, use_dense_prediction=True)   # main_esvit.py:306-310
        assert m.embed_dim == dim and m.blocks[0].attn.num_heads == heads and sum(p.numel() for p in m.parameters()) == params
        assert m.pos_embed.shape == (1, 197, dim) and m.blocks[-1].drop_path.drop_prob == pytest.approx(0.1)


def test_vit_composition_matches_reference_golden(cpu_ops):  # noqa: F811
    import esvit_amd.loss as L
    g = torch.load(os.path.join(GOLD, "nano_vit_step.pt"), weights_only=False)
    student, teacher = nano_vit_pair()
    s_out, t_out, loss = run_nano_vit_step(student, teacher, L)
    check_nano_vit(g, student, s_out, t_out, loss, rt=3e-4, loss_tol=2e-5, grad_tol=2e-3)


def test_vit_hooks_match_reference_golden(cpu_ops):  # noqa: F811
    g = torch.load(os.path.join(GOLD, "nano_vit_step.pt"), weights_only=False)
    student, _ = nano_vit_pair()
    check_nano_vit_hooks(g, student)


def test_vit_attention_restatement_matches_autograd():
    """oracle/ops_ref.vit_attn_fwd / _bwd (what the GPU kernels are compared with) against torch autograd of Attention.forward"""
    ops_ref.set_act_dtype(torch.float32)
    torch.manual_seed(0)
    B, N, nH, hd = 2, 37, 3, 16
    C = nH * hd
    qkv = torch.randn(B * N, 3 * C, requires_grad=True)
    x = qkv.view(B, N, 3, nH, hd).permute(2, 0, 3, 1, 4)
    a = ((x[0] @ x[1].transpose(-2, -1)) * hd ** -0.5).softmax(-1)
    y = (a @ x[2]).transpose(1, 2).reshape(B * N, C)
    gy = torch.randn_like(y)
    y.backward(gy)
    o, saved = ops_ref.vit_attn_fwd(qkv.detach(), B, N, nH, hd ** -0.5)
    d = ops_ref.vit_attn_bwd(gy, saved, B, N, nH, hd ** -0.5)
    assert torch.allclose(o, y, atol=1e-6) and torch.allclose(d, qkv.grad, atol=1e-6)


def test_vit_ragged_route_equals_per_group_schedule(cpu_ops):  # noqa: F811
    """all crops as rows of one matrix (the default) == one backbone pass per resolution group (the reference's schedule):
    outputs, loss and every gradient"""
    import esvit_amd.loss as L
    res = []
    for ragged in (True, False):
        student, teacher = nano_vit_pair()
        student.ragged_multi_crop = teacher.ragged_multi_crop = ragged
        s_out, t_out, loss = run_nano_vit_step(student, teacher, L)
        res.append((s_out, loss, {n: p.grad.clone() for n, p in student.named_parameters() if p.grad is not None}))
    (sa, la, ga), (sb, lb, gb) = res
    assert list(sa[3]) == list(sb[3]) and abs(la.item() - lb.item()) < 1e-6
    for a, b in zip(sa[:3], sb[:3]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    assert sorted(ga) == sorted(gb)
    for n in ga:
        assert torch.allclose(ga[n], gb[n], rtol=2e-4, atol=1e-7), n


def test_vit_drop_path_rows_share_the_sample_factor(cpu_ops):  # noqa: F811
    """the ragged route expands one DropPath draw per (block, branch, sample) to token rows: with a vanishing rate the factors are 1
    and the outputs equal the no-drop forward; with rate 1/2 a dropped sample's rows are the block input (both branches skipped)"""
    from esvit_amd.models.swin_transformer import DropPath
    student, _ = nano_vit_pair()
    crops = GU.make_crops(2, n_local=3, sizes=GU.NANO_VIT["sizes"])
    student.train()
    with torch.no_grad():
        want = student(crops)
        for blk in student.blocks:
            blk.drop_path = DropPath(1e-9)
        got = student(crops)
        for a, b in zip(want[:3], got[:3]):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
        f = student._drop_path_factors(10, torch.device("cpu"))
        assert f.shape == (GU.NANO_VIT["depth"], 2, 10) and torch.allclose(f, torch.ones_like(f), atol=1e-6)
        for blk in student.blocks:
            blk.drop_path = DropPath(0.5)
        student.__dict__.pop("_keep", None)
        torch.manual_seed(3)
        f = student._drop_path_factors(4000, torch.device("cpu"))
        assert set(f.unique().tolist()) == {0.0, 2.0} and abs(f.mean().item() - 1.0) < 0.05
        out = student(crops)   # runs through the row expansion with real drops
        assert all(torch.isfinite(t).all() for t in out[:3])


def test_vit_oracle_matches_reference_golden():
    """oracle/esvit_oracle.vit_multicrop (the functional restatement a GPU box can run at any size) against the step of the
    reference's own VisionTransformer + DINOHeads + DDINOLoss: outputs, loss, every gradient"""
    from oracle import esvit_oracle as O
    g = torch.load(os.path.join(GOLD, "nano_vit_step.pt"), weights_only=False)
    student, teacher = nano_vit_pair()
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in student.state_dict().items()}
    td = {k: v.detach().clone() for k, v in teacher.state_dict().items()}
    cfg = dict(depth=GU.NANO_VIT["depth"], heads=GU.NANO_VIT["heads"], patch=GU.NANO_VIT["patch"])
    crops = GU.make_crops(2, n_local=3, sizes=GU.NANO_VIT["sizes"])
    s_out = O.vit_multicrop(sd, crops, cfg)
    with torch.no_grad():
        t_out = O.vit_multicrop(td, crops[:2], cfg)
    assert (list(s_out[3]), list(t_out[3])) == g["npatch"]
    for nm, t in (("s_cls", s_out[0]), ("s_reg", s_out[1]), ("s_fea", s_out[2]), ("t_cls", t_out[0]), ("t_reg", t_out[1]), ("t_fea", t_out[2])):
        probe_close(nm, t.detach(), g[nm], rtol=2e-4)
    K = GU.NANO_HEAD["out_dim"]
    c0 = torch.zeros(1, K)
    loss, _, _ = O.ddino_loss(s_out, t_out, c0, c0, O.teacher_temp(2, 0.04, 0.07, 5, 10), 5)
    assert abs(loss.item() - g["ddino_loss"]) < 2e-5, (loss.item(), g["ddino_loss"])
    loss.backward()
    for n, ref in g["grad_norms"].items():
        got = sd[n].grad.norm().item()
        assert abs(got - ref) <= 2e-3 * ref + 1e-9, (n, got, ref)
    with torch.no_grad():
        probe_close("view-level", O.vit_multicrop(sd, crops, cfg, dense=False), g["view_only"], rtol=2e-4)


def test_vit_feature_extraction_for_knn(cpu_ops):  # noqa: F811
    """eval_knn.py:165-190 with a ViT backbone (eval_knn.py:115-118 builds it by name with num_classes = 0): extract_features returns
    the normed class tokens, row i at dataset index i -- equal to the functional oracle"""
    from functools import partial
    from esvit_amd import eval as E
    from esvit_amd.models import vision_transformer as V
    from oracle import esvit_oracle as O
    from tests.test_composition_cpu import IndexedSet
    v = GU.NANO_VIT
    model = V.VisionTransformer(img_size=[v["sizes"][0]], patch_size=v["patch"], embed_dim=v["embed_dim"], depth=v["depth"], num_heads=v["heads"],
                                mlp_ratio=4, qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_classes=0)
    GU.fill_state_dict(model.state_dict(), 5)
    model.eval()
    x = torch.randn(7, 3, v["sizes"][0], v["sizes"][0], generator=torch.Generator().manual_seed(4))
    loader = torch.utils.data.DataLoader(IndexedSet(x), batch_size=3)
    feats = E.extract_features(model, loader, use_cuda=False)
    sd = {k: t.detach() for k, t in model.state_dict().items()}
    with torch.no_grad():
        want = O.vit_features(sd, x, dict(depth=v["depth"], heads=v["heads"], patch=v["patch"]))[:, 0]
    assert feats.shape == want.shape and torch.allclose(feats, want, rtol=1e-4, atol=1e-5)
