"""where a step of the fused attention branch spends its cycles: per-phase cycle stamps of two waves of two workgroups
(tools/probe/build_ab_timeline.sh builds the instrumented library).  python tools/attn_branch_timeline.py [C H images shift]"""
import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("ESVIT_HIP_LIB", os.path.join(root, "tools", "probe", "libesvit_ab_timeline.so"))
sys.path.insert(0, root)
import torch
from esvit_amd import ops
from esvit_amd._lib import lib

C, H, nB, shift = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (96, 56, 256, 0)))
save = len(sys.argv) > 5 and sys.argv[5] == "save"
dev = torch.device("cuda:0")
ops.set_act_dtype(torch.bfloat16)
nH, ws, N, L = C // 32, 7, 49, H * H
w2t = torch.from_numpy(ops.window_maps(H, H, ws, shift)[0]).to(dev)
nW = w2t.numel() // N
reg = torch.from_numpy(ops.shift_region_ids(H, H, ws, shift)).to(dev) if shift else None
x = torch.randn(nB * L, C, device=dev)
g1, b1 = torch.ones(C, device=dev), torch.zeros(C, device=dev)
Wqkv, bqkv = torch.randn(3 * C, C, device=dev) * C ** -0.5, torch.randn(3 * C, device=dev) * 0.1
Wproj, bproj = torch.randn(C, C, device=dev) * C ** -0.5, torch.randn(C, device=dev) * 0.1
table = torch.randn((2 * ws - 1) ** 2, nH, device=dev) * 0.5
Wqp, Wpp = ops.cast_weight(Wqkv, perm32=True), ops.cast_weight(Wproj, perm32=True)
STEPS, NT = 48, 12
tl = torch.zeros(2 * STEPS * 2 * NT, dtype=torch.int32, device=dev)
fn = lib.esvit_attn_branch_timeline
fn.restype, fn.argtypes = None, [ctypes.c_void_p]
for rep in range(3):
    fn(ctypes.c_void_p(tl.data_ptr() if rep == 2 else 0))
    ops.attn_branch_fwd(x, g1, b1, 1e-6, Wqp, bqkv, Wpp, bproj, w2t, L, table, ws, reg, nW, N, nH, 32 ** -0.5, save=save)
torch.cuda.synchronize()
t = tl.cpu().view(2, STEPS, 2, NT).to(torch.int64) & 0xffffffff
names = ["issue", "K+S", "softmax", "PV", "proj", "stores+LN", "qkv", "late2+wait", "barrier", "(next top)"]
for blk in range(2):
    for wv in range(2):
        print("workgroup %d, %s wave: cycles per phase, steps 3.." % (blk, "first" if wv == 0 else "last"))
        acc = torch.zeros(NH := nH, 10)
        cnt = 0
        for st in range(nH, STEPS - 1):
            row = t[blk, st, wv]
            if row[9] == 0:
                break
            d = [(int(row[i + 1]) - int(row[i])) & 0xffffffff for i in range(9)] + [(int(t[blk, st + 1, wv, 0]) - int(row[9])) & 0xffffffff]
            acc[st % nH] += torch.tensor(d, dtype=torch.float32)
            if st % nH == nH - 1:
                cnt += 1
        for h in range(nH):
            print("  head step %d: " % h + "  ".join("%s %d" % (n, v) for n, v in zip(names, (acc[h] / max(cnt, 1)).tolist())) + "   total %d" % int(acc[h].sum() / max(cnt, 1)))
