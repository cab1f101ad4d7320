"""Wrap every esvit_amd.ops entry point: report the first call whose tensor inputs are all finite and whose output is not
(run on the MI355X):  python tools/probe/diag_nan_ops.py cvt_s1 16 [steps] [fwd]"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import esvit_amd
from esvit_amd import ops
from esvit_amd.engine import EsvitTrainer
from tests import golden_utils as GU

arch, B = sys.argv[1], int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
fwd_only = len(sys.argv) > 4
dev = torch.device("cuda:0")
esvit_amd.set_precision("bf16")
log, found = [], []


def flat(o):
    if torch.is_tensor(o):
        return [o]
    if isinstance(o, (tuple, list)):
        return [t for x in o for t in flat(x)]
    if isinstance(o, dict):
        return [t for x in o.values() for t in flat(x)]
    return []


def finite(ts):
    return all(bool(torch.isfinite(t.float()).all()) for t in ts if t.is_floating_point() and t.numel() and t.numel() < (1 << 31))


def wrap(name, fn):
    def w(*a, **k):
        ins = flat(a) + flat(k)
        ok_in = finite(ins)
        out = fn(*a, **k)
        ok_out = finite(flat(out)) and finite(ins)  # (in-place outputs are among the inputs)
        if ok_in and not ok_out and len(found) < 4:
            found.append((name, [(tuple(t.shape), str(t.dtype).replace("torch.", "")) for t in ins][:8], {kk: (vv if not torch.is_tensor(vv) else tuple(vv.shape)) for kk, vv in k.items()},
                          [(tuple(t.shape), bool(torch.isfinite(t.float()).all()) if t.is_floating_point() else None) for t in flat(out)][:6], len(log)))
        log.append((name, ok_in, ok_out))
        return out
    return w


SKIP = {"workspace", "query", "check", "gemm_select", "ops_module", "new_bias_frag"}
for n in dir(ops):
    f = getattr(ops, n)
    if isinstance(f, types.FunctionType) and not n.startswith("_") and n not in SKIP and f.__module__ == ops.__name__:
        setattr(ops, n, wrap(n, f))

torch.manual_seed(0)
student, teacher, loss_fn = bench.build(dev, 0.1, arch)
torch.manual_seed(1000)
trainer = EsvitTrainer(student, teacher, loss_fn, clip_grad=3.0, freeze_last_layer=1, teacher_stream=False)
crops = [c.to(dev) for c in GU.make_crops(B, seed=1234)]
lr, wd, mom, epoch = 5e-4 * B / 256.0, 0.04, 0.996, 1
if fwd_only:
    with torch.no_grad():
        teacher(crops[:2])
else:
    for i in range(steps):
        n0 = len(log)
        loss = trainer.step(crops, lr, wd, mom, epoch)
        print("step", i, "loss", loss.item(), "ops calls", len(log) - n0, flush=True)
        if found:
            break
for f in found:
    print("FIRST BAD:", f)
    i = f[-1]
    print("  calls before:", [l[0] for l in log[max(0, i - 6):i]])
if not found:
    print("no op with finite inputs and non-finite outputs; calls:", len(log))
