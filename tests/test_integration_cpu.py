"""Integration level 1 exercised for real: the reference's UNMODIFIED train_one_epoch
(main_esvit.py:499-600, compiled from /root/reference) drives OUR build_model / DINOHead / DDINOLoss objects with
torch.optim.AdamW over utils.get_params_groups, utils.clip_gradients, utils.cancel_gradients_last_layer, the positional EMA
zip and a DistributedDataParallel wrap -- and ends with the same parameters as the same loop over the reference's own
modules.  Then the checkpoint dict of main_esvit.py:476-488 is written and restored in both directions with the
reference's utils.save_on_master / restart_from_checkpoint.

CPU only (the GPU box has no /root/reference): the kernels are replaced by their torch restatement (oracle/ops_ref.py), the
host code under test is the product's.  The caller's fp32 branch references an undefined name (`model`, main_esvit.py:571)
whenever --clip_grad > 0, so the loop is driven through its scaler branch with the duck-typed no-op scaler documented in
SURVEY.md 8b hazard (i)."""
import argparse
import os

import pytest
import torch

from oracle import gen_golden as GG
from oracle import ref_loader as RL
from tests import golden_utils as GU
from tests.test_composition_cpu import cpu_ops, nano_pair  # noqa: F401  (fixture)

pytestmark = pytest.mark.skipif(not RL.available(), reason="needs the reference tree at /root/reference")


class NoOpScaler:
    """what a caller passes as fp16_scaler to take the (bug-free) scaler branch of main_esvit.py:576-584 without fp16"""

    def scale(self, loss):
        return loss

    def unscale_(self, optimizer):
        pass

    def step(self, optimizer):
        optimizer.step()

    def update(self):
        pass

    def state_dict(self):
        return {}

    def load_state_dict(self, sd):
        pass


class Loader:
    sampler = None

    def __init__(self, batches):
        self.batches = batches

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter([(b, None) for b in self.batches])


def _run(train_one_epoch, ns, student, teacher, loss_fn, batches, epoch, tmp):
    from torch.nn.parallel import DistributedDataParallel as DDP
    ddp = DDP(student)  # main_esvit.py:377 (gloo, world 1 here)
    opt = torch.optim.AdamW(ns.utils.get_params_groups(ddp))  # main_esvit.py:408-411
    n = len(batches)
    sched = dict(lr=[9.0] * n + [3e-4, 2e-4], wd=[9.0] * n + [0.04, 0.05], mom=[0.0] * n + [0.99, 0.995])
    args = argparse.Namespace(epochs=2, clip_grad=3.0, freeze_last_layer=1, batch_size_per_gpu=2, output_dir=str(tmp))
    stats = train_one_epoch(ddp, teacher, teacher, loss_fn, Loader(batches), opt, sched["lr"], sched["wd"], sched["mom"], epoch, None,
                            NoOpScaler(), args)
    return ddp, opt, stats, args


def test_level1_unmodified_train_one_epoch_and_checkpoints(cpu_ops, tmp_path, monkeypatch):  # noqa: F811
    import esvit_amd
    ns = RL.load()
    RL.ensure_single_process_group()
    train_one_epoch = RL.load_train_one_epoch()
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)  # main_esvit.py:513 on a CPU-only box
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)   # main_esvit.py:578, 593
    # the metric logger all-reduces its meters through a device='cuda' tensor (utils.py:223): single process here
    monkeypatch.setattr(ns.utils, "is_dist_avail_and_initialized", lambda: False)
    K = GU.NANO_HEAD["out_dim"]
    batches = [GU.make_crops(2, seed=70 + i) for i in range(2)]

    # (a) ours behind the unmodified loop
    student, teacher = nano_pair()
    loss_fn = esvit_amd.DDINOLoss(K, 10, 0.04, 0.07, 5, 10)
    ddp, opt, stats, args = _run(train_one_epoch, ns, student, teacher, loss_fn, batches, 1, tmp_path)
    assert set(stats) == {"loss", "lr", "wd"} and stats["loss"] == stats["loss"]

    # (b) the reference's own modules behind the same loop, same weights and crops
    r_student, r_teacher = GG.build_nano(ns), GG.build_nano(ns, teacher=True)
    GU.fill_state_dict(r_student.state_dict(), 0)
    GU.fill_state_dict(r_teacher.state_dict(), 7)
    r_student.head.last_layer.weight_g.data.fill_(1)
    for p in r_teacher.parameters():
        p.requires_grad = False
    r_loss = ns.DDINOLoss(K, 10, 0.04, 0.07, 5, 10)
    r_ddp, r_opt, r_stats, _ = _run(train_one_epoch, ns, r_student, r_teacher, r_loss, batches, 1, tmp_path)

    assert abs(stats["loss"] - r_stats["loss"]) < 1e-4, (stats, r_stats)
    # two AdamW steps: the first one moves every entry by ~lr * sign(g), so an entry whose gradient is ~0 can land a whole step
    # apart under fp32 round-off; what must agree is the update as a vector (relative L2 distance of the parameter deltas)
    init_s, init_t = nano_pair()
    init_s, init_t = dict(init_s.named_parameters()), dict(init_t.named_parameters())
    for (n, a), (_, b) in zip(ddp.module.named_parameters(), r_ddp.module.named_parameters()):
        ua, ub = (a - init_s[n]).detach(), (b - init_s[n]).detach()
        if ub.norm() > 0:
            assert (ua - ub).norm() <= 2e-2 * ub.norm(), (n, (ua - ub).norm().item(), ub.norm().item())
        else:
            assert torch.equal(a, b), n  # untouched on both sides (last layer during the freeze epoch would be; frozen weight_g)
    for (n, a), (_, b) in zip(teacher.named_parameters(), r_teacher.named_parameters()):
        ua, ub = (a - init_t[n]).detach(), (b - init_t[n]).detach()
        assert (ua - ub).norm() <= 1e-3 * ub.norm() + 1e-9, (n, (ua - ub).norm().item(), ub.norm().item())
    assert torch.allclose(loss_fn.center, r_loss.center, atol=1e-6) and torch.allclose(loss_fn.center_grid, r_loss.center_grid, atol=1e-6)
    sd, r_sd = opt.state_dict(), r_opt.state_dict()
    assert sd["param_groups"][0]["params"] == r_sd["param_groups"][0]["params"] and sorted(sd["state"]) == sorted(r_sd["state"])

    # (c) the checkpoint dict of main_esvit.py:476-488, written by one side and restored by the other, both directions
    def save(path, st, te, op, lf):
        ns.utils.save_on_master({"student": st.state_dict(), "teacher": te.state_dict(), "optimizer": op.state_dict(), "epoch": 2,
                                 "args": args, "dino_loss": lf.state_dict(), "fp16_scaler": {}}, path)

    def fresh_ours():
        from torch.nn.parallel import DistributedDataParallel as DDP
        s, t = nano_pair()
        for p in s.parameters():
            p.data.zero_()
        d = DDP(s)
        return d, t, torch.optim.AdamW(ns.utils.get_params_groups(d)), esvit_amd.DDINOLoss(K, 10, 0.04, 0.07, 5, 10)

    def fresh_ref():
        from torch.nn.parallel import DistributedDataParallel as DDP
        s, t = GG.build_nano(ns), GG.build_nano(ns, teacher=True)
        d = DDP(s)
        return d, t, torch.optim.AdamW(ns.utils.get_params_groups(d)), ns.DDINOLoss(K, 10, 0.04, 0.07, 5, 10)

    def same(sa, sb):
        assert list(sa.keys()) == list(sb.keys())
        for k in sa:
            assert torch.equal(sa[k], sb[k]), k

    ours_ckpt, ref_ckpt = os.path.join(tmp_path, "ours.pth"), os.path.join(tmp_path, "ref.pth")
    save(ours_ckpt, ddp, teacher, opt, loss_fn)
    save(ref_ckpt, r_ddp, r_teacher, r_opt, r_loss)
    # ours -> reference modules
    d, t, o, lf = fresh_ref()
    to_restore = {"epoch": 0}
    ns.utils.restart_from_checkpoint(ours_ckpt, run_variables=to_restore, student=d, teacher=t, optimizer=o, fp16_scaler=NoOpScaler(), dino_loss=lf)
    assert to_restore["epoch"] == 2
    same(d.state_dict(), ddp.state_dict())
    same(t.state_dict(), teacher.state_dict())
    same(lf.state_dict(), loss_fn.state_dict())
    assert sorted(o.state_dict()["state"]) == sorted(sd["state"])
    # reference -> our modules
    d, t, o, lf = fresh_ours()
    ns.utils.restart_from_checkpoint(ref_ckpt, run_variables=to_restore, student=d, teacher=t, optimizer=o, fp16_scaler=NoOpScaler(), dino_loss=lf)
    same(d.state_dict(), r_ddp.state_dict())
    same(t.state_dict(), r_teacher.state_dict())
    same(lf.state_dict(), r_loss.state_dict())
    k0 = sorted(r_sd["state"])[0]
    assert torch.equal(o.state_dict()["state"][k0]["exp_avg"], r_sd["state"][k0]["exp_avg"])


def test_level1_vit_behind_unmodified_train_one_epoch(cpu_ops, tmp_path, monkeypatch):  # noqa: F811
    """the same for the monolithic ViT (the reference's default --arch, main_esvit.py:305-327): our VisionTransformer + DINOHeads +
    DDINOLoss behind the reference's unmodified loop end with the loss and the parameter updates of the reference's own ViT"""
    import esvit_amd
    from tests.test_vit_cpu import nano_vit_pair
    ns = RL.load()
    RL.ensure_single_process_group()
    train_one_epoch = RL.load_train_one_epoch()
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(ns.utils, "is_dist_avail_and_initialized", lambda: False)
    K = GU.NANO_HEAD["out_dim"]
    batches = [GU.make_crops(2, n_local=3, sizes=GU.NANO_VIT["sizes"], seed=80 + i) for i in range(2)]
    student, teacher = nano_vit_pair()
    loss_fn = esvit_amd.DDINOLoss(K, 5, 0.04, 0.07, 5, 10)
    ddp, opt, stats, _ = _run(train_one_epoch, ns, student, teacher, loss_fn, batches, 1, tmp_path)
    r_student, r_teacher = GG.build_nano_vit(ns), GG.build_nano_vit(ns, teacher=True)
    GU.fill_state_dict(r_student.state_dict(), 0)
    GU.fill_state_dict(r_teacher.state_dict(), 7)
    r_student.head.last_layer.weight_g.data.fill_(1)
    for p in r_teacher.parameters():
        p.requires_grad = False
    r_loss = ns.DDINOLoss(K, 5, 0.04, 0.07, 5, 10)
    r_ddp, r_opt, r_stats, _ = _run(train_one_epoch, ns, r_student, r_teacher, r_loss, batches, 1, tmp_path)
    assert abs(stats["loss"] - r_stats["loss"]) < 1e-4, (stats, r_stats)
    init_s, init_t = nano_vit_pair()
    init_s, init_t = dict(init_s.named_parameters()), dict(init_t.named_parameters())
    assert [n for n, _ in ddp.module.named_parameters()] == [n for n, _ in r_ddp.module.named_parameters()]
    for (n, a), (_, b) in zip(ddp.module.named_parameters(), r_ddp.module.named_parameters()):
        ua, ub = (a - init_s[n]).detach(), (b - init_s[n]).detach()
        if ub.norm() > 0:
            assert (ua - ub).norm() <= 2e-2 * ub.norm(), (n, (ua - ub).norm().item(), ub.norm().item())
        else:
            assert torch.equal(a, b), n
    for (n, a), (_, b) in zip(teacher.named_parameters(), r_teacher.named_parameters()):
        ua, ub = (a - init_t[n]).detach(), (b - init_t[n]).detach()
        assert (ua - ub).norm() <= 1e-3 * ub.norm() + 1e-9, (n, (ua - ub).norm().item(), ub.norm().item())
    assert torch.allclose(loss_fn.center, r_loss.center, atol=1e-6) and torch.allclose(loss_fn.center_grid, r_loss.center_grid, atol=1e-6)
    # checkpoints are interchangeable: the reference's modules load ours
    r_fresh = GG.build_nano_vit(ns)
    r_fresh.load_state_dict(ddp.module.state_dict())
