"""Which host-side torch ops (copies, fills, adds) run inside one training step, and from where: torch.profiler with stacks."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import esvit_amd
from esvit_amd.engine import EsvitTrainer
from tests import golden_utils as GU
import bench
dev = torch.device("cuda:0")
esvit_amd.set_precision("bf16")
torch.manual_seed(0)
student, teacher, loss_fn = bench.build(dev, 0.1, sys.argv[1] if len(sys.argv) > 1 else "swin_tiny_w7")
tr = EsvitTrainer(student, teacher, loss_fn, clip_grad=3.0, freeze_last_layer=1)
crops = [c.to(dev) for c in GU.make_crops(8, seed=1)]
for _ in range(2):
    tr.step(crops, 5e-4, 0.04, 0.996, 1)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(crops, 5e-4, 0.04, 0.996, 1)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::clone", "aten::cat", "aten::mul", "aten::index", "aten::zeros", "aten::div_", "aten::mul_", "aten::floor_", "aten::uniform_", "aten::rand", "aten::to", "aten::_to_copy", "aten::contiguous", "aten::sub", "aten::abs_"):
        st = ev.stack or []
        inner = next((f for f in st if "esvit_amd" in f), None) or (st[0] if st else "engine/unknown")
        cnt[(ev.name, inner[-70:])] += 1
for k, v in cnt.most_common(40):
    print(v, k)
print("---- all ops by count ----")
allc = collections.Counter(ev.name for ev in prof.events())
for k, v in allc.most_common(45):
    print(v, k[:110])
