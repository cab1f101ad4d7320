"""Full-width step at bench sizes, bf16 mode against fp32 mode of the same library (mostly disjoint kernels): losses of the first steps
side by side (run on the MI355X):  python tools/probe/diag_prec.py swin_tiny_w7 32 [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import esvit_amd
from esvit_amd.engine import EsvitTrainer
from tests import golden_utils as GU

arch, B = sys.argv[1], int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda:0")
res = {}
for prec in ("fp32", "bf16"):
    esvit_amd.set_precision(prec)
    torch.manual_seed(0)
    student, teacher, loss_fn = bench.build(dev, 0.0, arch)
    torch.manual_seed(1000)
    trainer = EsvitTrainer(student, teacher, loss_fn, clip_grad=3.0, freeze_last_layer=1, teacher_stream=False)
    crops = [c.to(dev) for c in GU.make_crops(B, seed=1234)]
    lr, wd, mom, epoch = 5e-4 * B / 256.0, 0.04, 0.996, 1
    ls = []
    for i in range(steps):
        ls.append(trainer.step(crops, lr, wd, mom, epoch).item())
    gn = torch.sqrt(sum((p.detach().float() ** 2).sum() for p in student.parameters())).item()
    res[prec] = (ls, gn)
    del student, teacher, loss_fn, trainer, crops
    torch.cuda.empty_cache()
print(arch, B)
for prec, (ls, gn) in res.items():
    print("  %s losses %s  |params| %.6f" % (prec, ["%.5f" % v for v in ls], gn))
print("  max |loss diff| %.2e   rel param-norm diff %.2e" % (max(abs(a - b) for a, b in zip(res["fp32"][0], res["bf16"][0])),
      abs(res["fp32"][1] - res["bf16"][1]) / res["fp32"][1]))
