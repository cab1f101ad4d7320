"""per-step wall times (synchronised) of a bench configuration + allocator statistics (run on the MI355X):
    python tools/probe/diag_steptimes.py deit_tiny 128 16"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import esvit_amd
from esvit_amd.engine import EsvitTrainer
from tests import golden_utils as GU

arch, B, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
esvit_amd.set_precision("bf16")
torch.manual_seed(0)
student, teacher, loss_fn = bench.build(dev, 0.1, arch)
torch.manual_seed(1000)
trainer = EsvitTrainer(student, teacher, loss_fn, clip_grad=3.0, freeze_last_layer=1)
crops = [c.to(dev) for c in GU.make_crops(B, seed=1234)]
lr, wd, mom, epoch = 5e-4 * B / 256.0, 0.04, 0.996, 1
ts, segs, issue = [], [], []
for i in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    trainer.step(crops, lr, wd, mom, epoch)
    issue.append((time.perf_counter() - t0) * 1e3)  # the host has enqueued the step (the queue was empty when it started)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
    st = torch.cuda.memory_stats()
    segs.append((st["num_alloc_retries"], st["segment.all.allocated"], st["reserved_bytes.all.current"] >> 30, st["allocated_bytes.all.peak"] >> 30))
print(arch, B, "ms per step:", [round(t, 1) for t in ts])
print("  host issue time per step:", [round(t, 1) for t in issue], "cpu affinity", len(os.sched_getaffinity(0)), "cores")
print("  (alloc retries, segments allocated so far, reserved GiB, peak allocated GiB):", segs[0], segs[3], segs[-1])
