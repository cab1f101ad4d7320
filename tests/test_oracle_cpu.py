"""CPU tests (no GPU): the oracle restatement (oracle/esvit_oracle.py), the torch op restatement
(oracle/ops_ref.py) and the host-side integer maps of libesvit_hip.so against the committed golden
vectors generated from the reference's own modules (oracle/gen_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import esvit_oracle as O
from oracle import ops_ref
from tests import golden_utils as GU

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def maps():
    return np.load(os.path.join(GOLD, "index_maps.npz"))


@pytest.fixture(scope="module")
def nano():
    return torch.load(os.path.join(GOLD, "nano_step.pt"), weights_only=False)


def probe_close(name, t, p, rtol=2e-4):
    assert tuple(t.shape) == tuple(p["shape"]), (name, t.shape, p["shape"])
    q = GU.probe(t)
    scale = max(p["asum"] / max(1, t.numel()), 1e-12)
    for k in ("head", "strided"):
        err = (q[k] - p[k]).abs().max().item()
        ref = p[k].abs().max().item()
        assert err <= rtol * max(ref, scale), "%s.%s: err %.3e ref %.3e" % (name, k, err, ref)
    assert abs(q["asum"] - p["asum"]) <= rtol * p["asum"] + 1e-12, (name, q["asum"], p["asum"])


GEOMS = [(ws, H, s) for ws in (7, 14) for H in (56, 28, 14, 7, 24, 12, 6, 3) for s in (0, ws // 2)]


def test_index_maps_oracle_bit_exact(maps):
    for ws in (7, 14):
        assert np.array_equal(O.rel_pos_index(ws), maps["rpi_%d" % ws])
        assert np.array_equal(ops_ref.relative_position_index(ws), maps["rpi_%d" % ws])
    for ws, H, s in GEOMS:
        want = maps["win2tok_%d_%d_%d" % (ws, H, s)]
        assert np.array_equal(O.window_geometry(H, H, ws, s)[0].astype(np.int32), want), (ws, H, s)
        assert np.array_equal(ops_ref.window_maps(H, H, ws, s)[0], want), (ws, H, s)
        if s > 0:
            assert np.array_equal(O.shift_mask(H, H, ws, s), maps["mask_%d_%d" % (ws, H)]), (ws, H)
            assert np.array_equal(ops_ref.shift_mask(H, H, ws, s), maps["mask_%d_%d" % (ws, H)]), (ws, H)


def test_index_maps_library_bit_exact(maps, lib_built):
    """the C-ABI host functions esvit_relative_position_index / esvit_window_maps / esvit_shift_mask"""
    from esvit_amd import ops
    for ws in (7, 14):
        assert np.array_equal(ops.relative_position_index(ws), maps["rpi_%d" % ws])
    for ws, H, s in GEOMS:
        w2t, t2w = ops.window_maps(H, H, ws, s)
        assert np.array_equal(w2t, maps["win2tok_%d_%d_%d" % (ws, H, s)]), (ws, H, s)
        assert np.array_equal(w2t[t2w], np.arange(H * H)), "tok2win is not the inverse"
        if s > 0:
            assert np.array_equal(ops.shift_mask(H, H, ws, s), maps["mask_%d_%d" % (ws, H)]), (ws, H)
            ids = ops.shift_region_ids(H, H, ws, s).reshape(-1, ws * ws)
            rebuilt = np.where(ids[:, :, None] == ids[:, None, :], 0.0, -100.0).astype(np.float32)
            assert np.array_equal(rebuilt, maps["mask_%d_%d" % (ws, H)]), (ws, H)


def _nano_sd(nano, seed):
    sd = {k: torch.zeros(shape, dtype=getattr(torch, dt.split(".")[1])) for k, shape, dt in nano["keys"]}
    GU.fill_state_dict(sd, seed)
    for k in sd:
        if k.endswith("relative_position_index"):
            sd[k] = torch.as_tensor(O.rel_pos_index(int(round(sd[k].shape[0] ** 0.5))))
    return sd


def test_oracle_w14_matches_reference_golden():
    """14x14 windows (196 tokens, 27x27 bias table; stage 3 falls back to 7x7): BASELINE.json configs 3 and 4"""
    g14 = torch.load(os.path.join(GOLD, "nano14_step.pt"), weights_only=False)
    sd, tsd = _nano_sd(g14, 0), _nano_sd(g14, 7)
    sd["head.last_layer.weight_g"].fill_(1)
    names = [n for n in g14["grad_norms"]]
    params = {n: sd[n].clone().requires_grad_(True) for n in names}
    full = dict(sd)
    full.update(params)
    crops = GU.make_crops(1, n_local=2)
    s_out = O.swin_multicrop(full, crops, GU.NANO14)
    with torch.no_grad():
        t_out = O.swin_multicrop(tsd, crops[:2], GU.NANO14)
    for nm, t in (("s_cls", s_out[0]), ("s_reg", s_out[1]), ("s_fea", s_out[2]), ("t_cls", t_out[0]), ("t_reg", t_out[1])):
        probe_close(nm, t, g14[nm])
    K = GU.NANO_HEAD["out_dim"]
    loss, _, _ = O.ddino_loss(s_out, t_out, torch.zeros(1, K), torch.zeros(1, K), O.teacher_temp(2, 0.04, 0.07, 5, 10), 4)
    assert abs(loss.item() - g14["ddino_loss"]) < 1e-5
    loss.backward()
    for n, ref in g14["grad_norms"].items():
        assert abs(params[n].grad.norm().item() - ref) <= 1e-3 * ref + 1e-9, n
    for n, p in g14["grads"].items():
        probe_close("grad " + n, params[n].grad, p, rtol=1e-3)


def test_oracle_cvt_matches_reference_golden():
    """CvT (cvt_v4_transformer, BASELINE config 5): conv embeds, depthwise-conv + train-mode BatchNorm qkv, head_dim-64 windows
    (7x7, padded 12 -> 14, 6x6, 3x3), QuickGELU FFN -- outputs, loss and every gradient against the reference"""
    g = torch.load(os.path.join(GOLD, "nano_cvt_step.pt"), weights_only=False)
    shapes = {k: (shp, dt) for k, shp, dt in g["keys"]}

    def make_sd(seed):
        sd = {k: torch.zeros(shp, dtype=getattr(torch, dt.split(".")[1])) for k, (shp, dt) in shapes.items()}
        GU.fill_state_dict(sd, seed)
        return sd
    sd, tsd = make_sd(0), make_sd(7)
    sd["head.last_layer.weight_g"].fill_(1)
    names = list(g["grad_norms"])
    params = {n: sd[n].clone().requires_grad_(True) for n in names}
    full = dict(sd)
    full.update(params)
    crops = GU.make_crops(2, n_local=3, sizes=GU.NANO_CVT["sizes"])
    s_out = O.cvt_multicrop(full, crops, GU.NANO_CVT)
    with torch.no_grad():
        t_out = O.cvt_multicrop(tsd, crops[:2], GU.NANO_CVT)
    assert list(s_out[3]) == g["npatch"][0] and list(t_out[3]) == g["npatch"][1]
    for nm, t in (("s_cls", s_out[0]), ("s_reg", s_out[1]), ("s_fea", s_out[2]), ("t_cls", t_out[0]), ("t_reg", t_out[1]), ("t_fea", t_out[2])):
        probe_close(nm, t, g[nm])
    K = GU.NANO_HEAD["out_dim"]
    loss, _, _ = O.ddino_loss(s_out, t_out, torch.zeros(1, K), torch.zeros(1, K), O.teacher_temp(2, 0.04, 0.07, 5, 10), 5)
    assert abs(loss.item() - g["ddino_loss"]) < 1e-5
    loss.backward()
    for n, ref in g["grad_norms"].items():
        assert abs(params[n].grad.norm().item() - ref) <= 2e-3 * ref + 1e-9, n
    for n, p in g["grads"].items():
        probe_close("grad " + n, params[n].grad, p, rtol=2e-3)


def test_oracle_step_matches_reference_golden(nano):
    torch.manual_seed(0)
    K = GU.NANO_HEAD["out_dim"]
    sd = _nano_sd(nano, 0)
    sd["head.last_layer.weight_g"].fill_(1)
    tsd = _nano_sd(nano, 7)
    params = {n: sd[n].clone().requires_grad_(n in nano["trainable"]) for n in nano["param_names"]}
    full = dict(sd)
    full.update(params)
    crops = GU.make_crops(2)
    s_out = O.swin_multicrop(full, crops, GU.NANO)
    with torch.no_grad():
        t_out = O.swin_multicrop(tsd, crops[:2], GU.NANO)
    for nm, t, key in (("s_cls", s_out[0], "s_cls"), ("s_reg", s_out[1], "s_reg"), ("s_fea", s_out[2], "s_fea"),
                       ("t_cls", t_out[0], "t_cls"), ("t_reg", t_out[1], "t_reg"), ("t_fea", t_out[2], "t_fea")):
        probe_close(nm, t, nano[key])
    assert (list(s_out[3]), list(t_out[3])) == nano["npatch"]
    assert (s_out[0] - nano["s_cls_full"]).abs().max().item() < 2e-5
    assert (s_out[1][::17] - nano["s_reg_rows"]).abs().max().item() < 2e-5
    temp = O.teacher_temp(2, 0.04, 0.07, 5, 10)
    loss, bc, bg = O.ddino_loss(s_out, t_out, nano["center0"], nano["center_grid0"], temp, 10)
    assert abs(loss.item() - nano["ddino_loss"]) < 1e-5
    c1 = O.center_update(nano["center0"], bc, t_out[0].shape[0])
    cg1 = O.center_update(nano["center_grid0"], bg, t_out[1].shape[0])
    assert (c1 - nano["center1"]).abs().max().item() < 1e-6 and (cg1 - nano["center_grid1"]).abs().max().item() < 1e-6
    loss.backward()
    assert [n for n in nano["param_names"] if params[n].grad is None] == nano["no_grad"]
    for n, p in nano["grads"].items():
        probe_close("grad " + n, params[n].grad, p, rtol=1e-3)
        assert abs(params[n].grad.norm().item() - nano["grad_norms"][n]) <= 1e-3 * nano["grad_norms"][n] + 1e-9
    with torch.no_grad():
        l2, _, _ = O.ddino_loss([t.detach() if torch.is_tensor(t) else t for t in s_out], t_out, c1, cg1, temp, 10)
        assert abs(l2.item() - nano["ddino_loss_2"]) < 1e-5
        cls2 = O.dino_head(sd, "head.", O.swin_features(sd, torch.cat(crops[:2]), GU.NANO)[0])
        lv, bcv = O.dino_loss(cls2, t_out[0], nano["center0"], temp, 2)
        assert abs(lv.item() - nano["dino_loss_2crops"]) < 1e-5
        assert (O.center_update(nano["center0"], bcv, t_out[0].shape[0]) - nano["dino_center1"]).abs().max().item() < 1e-6
    # update step
    pd = {n: params[n].detach().clone() for n in nano["param_names"]}
    gd = {n: params[n].grad for n in nano["param_names"] if params[n].grad is not None}
    td = {n: tsd[n].clone() for n in nano["param_names"]}
    reg = {n for n in nano["trainable"] if not (n.endswith(".bias") or pd[n].ndim == 1)}
    assert [len(reg), len(nano["trainable"]) - len(reg)] == nano["group_sizes"]
    O.clip_adamw_ema(pd, gd, {}, td, reg, 5e-4, 0.04, 0.996, clip=3.0)
    for n in nano["param_names"]:
        probe_close("student_after " + n, pd[n], nano["student_after"][n], rtol=1e-4)
        probe_close("teacher_after " + n, td[n], nano["teacher_after"][n], rtol=1e-4)
    with torch.no_grad():
        _, _, attns = O.swin_features(sd, crops[0], GU.NANO, return_attn=True)
        probe_close("last_attn", attns[-1], nano["last_attn"])


def test_oracle_knn_matches_reference_golden():
    """oracle.knn_classifier vs the reference's own knn_classifier (eval_knn.py:193-232) on the committed synthetic sets"""
    gold = torch.load(os.path.join(GOLD, "knn.pt"), weights_only=False)
    assert gold["cases"] == GU.KNN_CASES
    for c, want in zip(GU.KNN_CASES, gold["top"]):
        xtr, ytr, xte, yte = GU.make_knn_set(c["seed"], noise=c["noise"])
        got = O.knn_classifier(xtr, ytr, xte, yte, c["k"], c["T"], num_classes=10)
        assert got == pytest.approx(want, abs=1e-9), (c, got, want)


def test_lib_exports_every_declared_symbol(lib_built):
    """the C-ABI library loads on a GPU-less host and exports exactly what include/esvit_hip.h declares; the ctypes
    binding (esvit_amd/_lib.py) covers every one of them -- no compute is called"""
    import ctypes
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "esvit_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(esvit_[a-z0-9_]+)\s*\(", hdr))
    assert 40 <= len(declared) <= 60, len(declared)  # the ABI is meant to stay this small (the step, the crop producer, ViT / ViL attention: 57 in round 3, 59 with the wide MLP's training entry)
    lib = ctypes.CDLL(lib_built)
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    from esvit_amd import _lib
    unbound = sorted(declared - set(_lib.SIGNATURES))
    assert not unbound, unbound
    undeclared = sorted(set(_lib.SIGNATURES) - declared)
    assert not undeclared, undeclared
    assert _lib.lib.esvit_version() > 0
