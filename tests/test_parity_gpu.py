"""-m gpu: the parity holes VERDICT round 1 listed, closed on the HIP path.

* BASELINE config 1 (2 x 224^2 crops, view-level DINOLoss, use_dense_prediction=False): nano model vs the reference golden,
  Swin-T widths at bs 4 and out_dim 65536 vs the CPU oracle (main_esvit.py:603-660, swin_transformer.py:753-763);
* like-for-like bf16: the HIP bf16 step vs the torch restatement of the same ops with bf16 storage at the same rounding
  points (oracle/ops_ref.py, set_act_dtype(bf16)) -- catches a bug that only the bf16 kernels have -- with the observed
  HIP-bf16 vs fp32-reference deltas printed;
* cancel_gradients_last_layer (utils.py:118-123): the fused update with skip_last_layer=True vs the reference's
  clip -> cancel -> AdamW -> EMA result;
* Swin forward_return_n_last_blocks (swin_transformer.py:799-837) vs the reference golden;
* one Swin-T step at out_dim 65536, B = 8 vs the CPU oracle (loss + sampled gradient tensors)."""
import os

import pytest
import torch

from oracle import esvit_oracle as O
from oracle import ops_ref
from tests import golden_utils as GU
from tests.test_composition_cpu import build_nano, build_nano_view, nano_pair, run_nano_step
from tests.test_oracle_cpu import GOLD, probe_close

pytestmark = pytest.mark.gpu

_record = GU.record_parity


@pytest.fixture(scope="module")
def nano():
    return torch.load(os.path.join(GOLD, "nano_step.pt"), weights_only=False)


def _setup(prec):
    import esvit_amd
    assert torch.cuda.is_available()
    esvit_amd.set_precision(prec)
    return torch.device("cuda:0")


def _teardown():
    import esvit_amd
    esvit_amd.set_precision("bf16")


def _rel(a, b):
    return ((a.float().cpu() - b).abs().max() / (b.abs().max() + 1e-12)).item()


# ---------------------------------------------------------------------------------------------------------------------------
# BASELINE config 1
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_config1_nano_matches_reference_golden(nano, prec, lib_built):
    """use_dense_prediction=False model + DINOLoss(ncrops=2) on the two global crops: loss and centre vs the reference's
    own DINOLoss (golden dino_loss_2crops / dino_center1), every parameter receives a gradient, teacher untouched"""
    import esvit_amd
    dev = _setup(prec)
    try:
        student, teacher = build_nano_view(), build_nano_view(teacher=True)
        GU.fill_state_dict(student.state_dict(), 0)
        GU.fill_state_dict(teacher.state_dict(), 7)
        student.head.last_layer.weight_g.data.fill_(1)
        for p in teacher.parameters():
            p.requires_grad = False
        student, teacher = student.to(dev), teacher.to(dev)
        crops = [c.to(dev) for c in GU.make_crops(2)[:2]]
        K = GU.NANO_HEAD["out_dim"]
        loss_fn = esvit_amd.DINOLoss(K, 2, 0.04, 0.07, 5, 10).to(dev)
        loss_fn.center.copy_(nano["center0"].to(dev))
        with torch.no_grad():
            t_out = teacher(crops)
        s_out = student(crops)
        assert torch.is_tensor(s_out) and s_out.shape == (4, K) and t_out.shape == (4, K)
        probe_close("t_cls (view-only forward)", t_out.float().cpu(), nano["t_cls"], rtol=3e-4 if prec == "fp32" else 3e-2)
        loss = loss_fn(s_out, t_out, 2, None)
        loss.backward()
        fp = prec == "fp32"
        d = abs(loss.item() - nano["dino_loss_2crops"])
        _record(test="config1_nano", prec=prec, loss=loss.item(), ref=nano["dino_loss_2crops"], abs_err=d)
        assert d < (1e-4 if fp else 5e-3), (loss.item(), nano["dino_loss_2crops"])
        assert (loss_fn.center.cpu() - nano["dino_center1"]).abs().max().item() < (1e-6 if fp else 1e-3)
        missing = [n for n, p in student.named_parameters() if p.requires_grad and p.grad is None]
        assert missing == [], missing
    finally:
        _teardown()


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_config1_swin_tiny_bs4_matches_reference_golden(prec, lib_built):
    """BASELINE.json configs[0] exactly: Swin-T W=7, 2 global 224^2 crops only, view-level loss, bs 4, out_dim 65536 -- HIP
    path vs the step of the reference's own modules on the same weights and crops (tests/golden/full_width.pt): logits,
    loss, centre, every gradient norm, sampled gradient tensors"""
    from tests.test_step_gpu import FULL_GOLD, full_case_deltas, run_full_case
    g = torch.load(FULL_GOLD, map_location="cpu", weights_only=False)["config1_bs4"]
    dev = _setup(prec)
    try:
        student, loss_fn, s_out, t_out, loss = run_full_case("config1_bs4", dev)
        fp = prec == "fp32"
        out_rel, norm_rel, worst, worst_name = full_case_deltas(g, student, s_out)
        t_rel = ((GU.strided(t_out).float().cpu() - g["t_out"]).abs().max() / g["t_out"].abs().max()).item()
        c_err = (loss_fn.center.cpu() - g["center"]).abs().max().item()
        _record(test="config1_swin_tiny_bs4", prec=prec, loss=loss.item(), ref=g["loss"], abs_err=abs(loss.item() - g["loss"]),
                logits_rel=out_rel, teacher_logits_rel=t_rel, center_abs=c_err, worst_grad_norm_rel=norm_rel, worst_sampled_grad_rel_l2=worst,
                worst_tensor=worst_name)
        # bf16 bounds: <= 3x the deltas observed on MI355X (profiles/r03_parity_observed.jsonl: loss 8.9e-5, logits 6.9e-3 / 8.5e-3,
        # centre 1.5e-4, gradient norms 6.9e-3, sampled gradient tensors 2.2e-2 relative L2)
        assert out_rel < (1e-4 if fp else 2e-2) and t_rel < (1e-4 if fp else 2.5e-2), (out_rel, t_rel)
        assert abs(loss.item() - g["loss"]) < (1e-4 if fp else 2.7e-4), (loss.item(), g["loss"])
        assert c_err < (1e-6 if fp else 4.5e-4)
        assert norm_rel < (5e-3 if fp else 2e-2), norm_rel
        assert worst < (5e-3 if fp else 7e-2), (worst_name, worst)
    finally:
        _teardown()


# ---------------------------------------------------------------------------------------------------------------------------
# like-for-like bf16
# ---------------------------------------------------------------------------------------------------------------------------
def _emulated_bf16_nano_step(nano, monkeypatch):
    """the nano step through the product's host code with every op replaced by its torch restatement storing activations
    in bf16 at the same points the HIP kernels do (CPU)"""
    import esvit_amd.functional as Fn
    import esvit_amd.loss as L
    import esvit_amd.params as P
    with monkeypatch.context() as m:
        for mod in (Fn, L, P):
            m.setattr(mod, "ops", ops_ref)
        ops_ref.set_act_dtype(torch.bfloat16)
        P.clear()
        Fn._GEOM.clear()
        try:
            student, teacher = nano_pair()
            s_out, t_out, loss, _ = run_nano_step(nano, student, teacher, L, GU.make_crops(2))
            grads = {n: p.grad.detach().clone() for n, p in student.named_parameters() if p.grad is not None}
            out = (loss.item(), [t.detach().float().clone() for t in s_out[:3]], grads)
        finally:
            ops_ref.set_act_dtype(torch.float32)
            P.clear()
            Fn._GEOM.clear()
    return out


def test_bf16_step_like_for_like(nano, monkeypatch, lib_built):
    """HIP bf16 step vs the bf16-storage emulation of the same op sequence (a second, independent bf16 implementation: the
    emulation rounds every stored activation to bf16 where the kernels do, but multiplies in fp32 and rounds the softmax
    / CE intermediates at slightly different places) and vs the fp32 reference golden.  A bug that only the bf16 kernels
    have shows up as an O(1) relative error in some gradient tensor; bf16 rounding noise stays below the bounds asserted
    here (observed on MI355X, round 2: loss 2.0e-3 from the emulation and 7.5e-4 from the fp32 reference -- the HIP path
    is closer to fp32 than the emulation is --, worst gradient tensor 0.14 relative L2 from the emulation, 0.8 % in norm
    from the fp32 reference).  The observed deltas are printed."""
    import esvit_amd.loss as L
    l_emu, s_emu, g_emu = _emulated_bf16_nano_step(nano, monkeypatch)
    dev = _setup("bf16")
    try:
        student, teacher = nano_pair()
        student, teacher = student.to(dev), teacher.to(dev)
        nano_dev = dict(nano)
        nano_dev["center0"], nano_dev["center_grid0"] = nano["center0"].to(dev), nano["center_grid0"].to(dev)
        s_out, t_out, loss, _ = run_nano_step(nano_dev, student, teacher, L, [c.to(dev) for c in GU.make_crops(2)], dev=dev)
        worst_l2, worst_name, worst_vs_fp32 = 0.0, "", 0.0
        for n, p in student.named_parameters():
            if p.grad is None:
                continue
            g, e = p.grad.float().cpu(), g_emu[n]
            d = ((g - e).norm() / (e.norm() + 1e-12)).item()
            if d > worst_l2:
                worst_l2, worst_name = d, n
            ref = nano["grad_norms"][n]
            worst_vs_fp32 = max(worst_vs_fp32, abs(g.norm().item() - ref) / (ref + 1e-12))
        logit_rel = max(_rel(a, b) for a, b in zip(s_out[:3], s_emu))
        _record(test="bf16_like_for_like_nano", loss_hip=loss.item(), loss_emulated=l_emu, loss_fp32_reference=nano["ddino_loss"],
                hip_vs_emulated=abs(loss.item() - l_emu), hip_vs_fp32_reference=abs(loss.item() - nano["ddino_loss"]),
                worst_grad_rel_l2_vs_emulated=worst_l2, worst_grad_tensor=worst_name, worst_grad_norm_rel_vs_fp32_reference=worst_vs_fp32,
                logits_rel_vs_emulated=logit_rel)
        # <= 3x observed (round 2: 2.0e-3, 7.5e-4, 9.3e-3, 0.138, 8.2e-3)
        assert abs(loss.item() - l_emu) < 5e-3, (loss.item(), l_emu)
        assert abs(loss.item() - nano["ddino_loss"]) < 2.2e-3, (loss.item(), nano["ddino_loss"])
        assert logit_rel < 2.8e-2, logit_rel
        assert worst_l2 < 0.25, (worst_name, worst_l2)
        assert worst_vs_fp32 < 2.5e-2, worst_vs_fp32
    finally:
        _teardown()


# ---------------------------------------------------------------------------------------------------------------------------
# cancel_gradients_last_layer
# ---------------------------------------------------------------------------------------------------------------------------
def test_cancel_last_layer_update_matches_reference_golden(nano, lib_built):
    """epoch < freeze_last_layer: clip -> cancel_gradients_last_layer -> AdamW -> EMA (main_esvit.py:569-590, utils.py:118-123).
    The fused update with skip_last_layer=True must leave every `last_layer` parameter of the student untouched (no
    weight decay, no moments, no step count), update the rest exactly as without the flag, and EMA every teacher tensor."""
    import esvit_amd.loss as L
    from esvit_amd.update import FusedClipAdamWEMA
    dev = _setup("fp32")
    try:
        student, teacher = nano_pair()
        student, teacher = student.to(dev), teacher.to(dev)
        before = {n: p.detach().clone() for n, p in student.named_parameters()}
        nano_dev = dict(nano)
        nano_dev["center0"], nano_dev["center_grid0"] = nano["center0"].to(dev), nano["center_grid0"].to(dev)
        run_nano_step(nano_dev, student, teacher, L, [c.to(dev) for c in GU.make_crops(2)], dev=dev)
        upd = FusedClipAdamWEMA(student, teacher)
        upd.step(5e-4, 0.04, 0.996, clip_grad=3.0, skip_last_layer=True)
        torch.cuda.synchronize()
        assert nano["cancelled"] == [n for n in before if "last_layer" in n and n not in nano["no_grad"]]
        for n, p in student.named_parameters():
            probe_close("student_after_cancel " + n, p.detach().cpu(), nano["student_after_cancel"][n], rtol=2e-4)
            if "last_layer" in n:
                assert torch.equal(p.detach(), before[n]), n
        for n, p in teacher.named_parameters():
            probe_close("teacher_after_cancel " + n, p.detach().cpu(), nano["teacher_after_cancel"][n], rtol=2e-4)
        sd = upd.state_dict()
        names = [n for g in ("regularized", "not") for n in []]  # (state is keyed by position; count the stepped tensors instead)
        del names
        stepped = len(sd["state"])
        trainable_with_grad = sum(1 for n, p in student.named_parameters() if p.requires_grad and n not in nano["no_grad"])
        assert stepped == trainable_with_grad - len(nano["cancelled"]), (stepped, trainable_with_grad, nano["cancelled"])
    finally:
        _teardown()


# ---------------------------------------------------------------------------------------------------------------------------
# eval_linear.py's feature hook on the Swin backbone
# ---------------------------------------------------------------------------------------------------------------------------
def test_swin_forward_return_n_last_blocks_matches_reference_golden(nano, lib_built):
    dev = _setup("fp32")
    try:
        student = build_nano()
        GU.fill_state_dict(student.state_dict(), 0)
        student = student.to(dev).eval()
        crops = [c.to(dev) for c in GU.make_crops(2)]
        depth = list(GU.NANO["depths"])
        with torch.no_grad():
            f3 = student.forward_return_n_last_blocks(crops[0], n=3, depth=depth)
            f1 = student.forward_return_n_last_blocks(crops[2], n=1, depth=depth)
        assert f3.shape == nano["last_blocks_n3"].shape and f1.shape == nano["last_blocks_n1_local"].shape
        assert torch.allclose(f3.cpu(), nano["last_blocks_n3"], rtol=3e-4, atol=2e-5), (f3.cpu() - nano["last_blocks_n3"]).abs().max()
        assert torch.allclose(f1.cpu(), nano["last_blocks_n1_local"], rtol=3e-4, atol=2e-5)
    finally:
        _teardown()


# ---------------------------------------------------------------------------------------------------------------------------
# the benchmark's own configuration at a size the CPU oracle still finishes: out_dim 65536, B = 8
# ---------------------------------------------------------------------------------------------------------------------------
def test_swin_tiny_k65536_b8_step_matches_reference_golden(lib_built):
    """Swin-T W=7, 2x224^2 + 8x96^2 crops, DDINOLoss, out_dim 65536, B = 8 per GPU, bf16 (the precision bench.py times):
    loss, every gradient norm and twelve sampled gradient tensors against the fp32 step of the reference's own modules
    (tests/golden/full_width.pt); drop_path 0 so both sides are deterministic"""
    from tests.test_step_gpu import FULL_GOLD, full_case_deltas, run_full_case
    g = torch.load(FULL_GOLD, map_location="cpu", weights_only=False)["swin_t_k65536_b8"]
    dev = _setup("bf16")
    try:
        student, loss_fn, s_out, t_out, loss = run_full_case("swin_t_k65536_b8", dev)
        out_rel, norm_rel, worst, worst_name = full_case_deltas(g, student, s_out)
        _record(test="swin_tiny_k65536_b8", loss_hip_bf16=loss.item(), loss_reference_fp32=g["loss"], abs_err=abs(loss.item() - g["loss"]),
                outputs_rel=out_rel, worst_grad_norm_rel=norm_rel, worst_sampled_grad_rel_l2=worst, worst_tensor=worst_name)
        # <= 3x observed (round 3: loss 5.8e-5, outputs 8.2e-3, gradient norms 1.0e-2, sampled gradient tensors 5.2e-2)
        assert abs(loss.item() - g["loss"]) < 1.7e-4, (loss.item(), g["loss"])
        assert out_rel < 2.5e-2, out_rel
        assert norm_rel < 3e-2, norm_rel
        assert worst < 0.1, (worst_name, worst)
        assert (loss_fn.center.cpu() - g["center"]).abs().max().item() < 2e-3
    finally:
        _teardown()
