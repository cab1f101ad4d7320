"""First module whose output goes non-finite in a no-grad forward of a bench configuration (run on the MI355X):
    python tools/probe/diag_nan_fwd.py cvt_s1 16"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import esvit_amd
from tests import golden_utils as GU

arch, B = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda:0")
esvit_amd.set_precision(sys.argv[3] if len(sys.argv) > 3 else "bf16")
torch.manual_seed(0)
student, teacher, loss_fn = bench.build(dev, 0.1, arch)
crops = [c.to(dev) for c in GU.make_crops(B, seed=1234)]
print("crops", [(tuple(c.shape), c.dtype, float(c.abs().max())) for c in crops[:3]])
bad_params = [n for n, p in teacher.state_dict().items() if torch.is_floating_point(p) and not bool(torch.isfinite(p).all())]
print("non-finite teacher state:", bad_params[:10], len(bad_params))
seen = []


def flat(o):
    if torch.is_tensor(o):
        return [o]
    if isinstance(o, (tuple, list)):
        return [t for x in o for t in flat(x)]
    return []


def hook(name):
    def h(mod, inp, out):
        ok_in = all(bool(torch.isfinite(t.float()).all()) for t in flat(inp) if t.is_floating_point())
        ok = all(bool(torch.isfinite(t.float()).all()) for t in flat(out) if t.is_floating_point())
        mx = max([float(t.float().abs().max()) for t in flat(out) if t.is_floating_point() and t.numel()] or [0.0])
        seen.append((name, type(mod).__name__, ok_in, ok, mx, [tuple(t.shape) for t in flat(out)][:2]))
    return h


for n, m in teacher.named_modules():
    if n:
        m.register_forward_hook(hook(n))
with torch.no_grad():
    out = teacher(crops[:2])
first = True
for s in seen:
    if not s[3] and first:
        print("FIRST NON-FINITE OUTPUT:", s)
        first = False
for s in seen[:60]:
    print(s)
