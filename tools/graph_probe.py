"""Feasibility probe: capture one whole training step (teacher fwd, student fwd, loss, backward, fused update) in a hipGraph through
torch.cuda.graph and replay it -- python tools/graph_probe.py [arch] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import esvit_amd
from esvit_amd.engine import EsvitTrainer
import bench
arch = sys.argv[1] if len(sys.argv) > 1 else "swin_tiny_w7"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
esvit_amd.set_precision("bf16")
torch.manual_seed(0)
student, teacher, loss_fn = bench.build(dev, 0.1, arch)
tr = EsvitTrainer(student, teacher, loss_fn, clip_grad=3.0, freeze_last_layer=0, teacher_stream=False)
crops = [torch.randn(B, 3, 224, 224, device=dev) for _ in range(2)] + [torch.randn(B, 3, 96, 96, device=dev) for _ in range(8)]
def step():
    return tr.step(crops, 5e-4, 0.04, 0.996, 1)
for _ in range(3):
    l = step()
torch.cuda.synchronize()
def timeit(fn, n=10):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print("eager ms/step", round(timeit(step), 3), "loss", l.item())
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    lg = step()
torch.cuda.synchronize()
print("captured")
print("graph ms/step", round(timeit(g.replay), 3), "loss", lg.item())
