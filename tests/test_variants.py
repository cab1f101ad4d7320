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
