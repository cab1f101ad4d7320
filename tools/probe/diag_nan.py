"""Which step / which tensor of a non-Swin bench configuration first goes non-finite (run on the MI355X):
    python tools/probe/diag_nan.py deit_tiny 128 [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import esvit_amd
from esvit_amd.engine import EsvitTrainer
from tests import golden_utils as GU

arch, B = sys.argv[1], int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = torch.device("cuda:0")
esvit_amd.set_precision("bf16")
torch.manual_seed(0)
student, teacher, loss_fn = bench.build(dev, 0.1, arch)
torch.manual_seed(1000)
trainer = EsvitTrainer(student, teacher, loss_fn, clip_grad=3.0, freeze_last_layer=1, teacher_stream=False)
crops = [c.to(dev) for c in GU.make_crops(B, seed=1234)]
lr, wd, mom, epoch = 5e-4 * B / 256.0, 0.04, 0.996, 1


def fin(t):
    if torch.is_tensor(t):
        f = t.float()
        return "%s finite=%s absmax=%.4g" % (tuple(t.shape), bool(torch.isfinite(f).all()), f[torch.isfinite(f)].abs().max().item() if torch.isfinite(f).any() else float("nan"))
    if isinstance(t, (tuple, list)):
        return "[" + "; ".join(fin(x) for x in t) + "]"
    return str(type(t))


with torch.no_grad():
    print(arch, B, "teacher out:", fin(teacher(crops[:2])), flush=True)
    print(arch, B, "student out:", fin(student(crops)), flush=True)
for i in range(steps):
    loss = trainer.step(crops, lr, wd, mom, epoch)
    bad = [n for n, p in student.named_parameters() if not bool(torch.isfinite(p).all())]
    print("step", i, "loss", loss.item(), "skipped", trainer.updater.take_skipped() if hasattr(trainer.updater, "take_skipped") else None,
          "non-finite params:", bad[:6], len(bad), "centre finite", bool(torch.isfinite(loss_fn.center).all()), flush=True)
