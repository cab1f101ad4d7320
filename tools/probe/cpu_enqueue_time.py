"""How long the host needs to ENQUEUE one training step (no synchronisation inside the loop) against the step's GPU time: if the two are close the
step is launch-bound in places.  python tools/probe/cpu_enqueue_time.py [arch] [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import esvit_amd  # noqa: E402
from esvit_amd.data import synthetic_crops  # noqa: E402
from esvit_amd.engine import EsvitTrainer  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "swin_tiny_w7"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda", 0)
esvit_amd.set_precision("bf16")
torch.manual_seed(0)
student, teacher, loss_fn = bench.build(dev, 0.1, arch)
trainer = EsvitTrainer(student, teacher, loss_fn, clip_grad=3.0, freeze_last_layer=1)
crops = [c.to(dev) for c in synthetic_crops(B, seed=1234)]
lr, wd, mom = 5e-4 * B / 256.0, 0.04, 0.996
for _ in range(5):
    trainer.step(crops, lr, wd, mom, 1)
torch.cuda.synchronize()
N = 20
host = []
t0 = time.perf_counter()
for _ in range(N):
    a = time.perf_counter()
    trainer.step(crops, lr, wd, mom, 1)
    host.append(time.perf_counter() - a)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
host.sort()
print("arch %s B %d: step %.2f ms; host loop returned after %.2f ms per step (min %.2f, median %.2f of the per-call times)" %
      (arch, B, t_all / N * 1e3, t_enq / N * 1e3, host[0] * 1e3, host[N // 2] * 1e3))
