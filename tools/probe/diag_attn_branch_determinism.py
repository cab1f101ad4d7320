"""run-to-run bit equality of the fused attention branch at bench occupancy (the butterfly reductions are hand-written asm):
python tools/probe/diag_attn_branch_determinism.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from esvit_amd import ops
dev = torch.device("cuda:0")
ops.set_act_dtype(torch.bfloat16)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for (C, nH, H, nB, shift) in ((96, 3, 56, 64, 3), (96, 3, 24, 256, 0), (192, 6, 28, 64, 0), (192, 6, 12, 256, 3)):
    ws, N, L = 7, 49, H * H
    w2t = torch.from_numpy(ops.window_maps(H, H, ws, shift)[0]).to(dev)
    nW = w2t.numel() // N
    reg = torch.from_numpy(ops.shift_region_ids(H, H, ws, shift)).to(dev) if shift else None
    g = torch.Generator().manual_seed(3)
    x = torch.randn(nB * L, C, generator=g).to(dev)
    g1, b1 = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    Wqkv, bqkv = (torch.randn(3 * C, C, generator=g) * C ** -0.5).to(dev), (torch.randn(3 * C, generator=g) * 0.1).to(dev)
    Wproj, bproj = (torch.randn(C, C, generator=g) * C ** -0.5).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
    table = (torch.randn(169, nH, generator=g) * 0.5).to(dev)
    Wqp, Wpp = ops.cast_weight(Wqkv, perm32=True), ops.cast_weight(Wproj, perm32=True)
    ref, bad = None, 0
    for rep in range(reps):
        y, sv = ops.attn_branch_fwd(x, g1, b1, 1e-6, Wqp, bqkv, Wpp, bproj, w2t, L, table, ws, reg, nW, N, nH, 32 ** -0.5, save=True)
        y2 = ops.attn_branch_fwd(x, g1, b1, 1e-6, Wqp, bqkv, Wpp, bproj, w2t, L, table, ws, reg, nW, N, nH, 32 ** -0.5)
        cur = [y.clone(), y2.clone()] + [t.clone() for t in sv]
        if ref is None:
            ref = cur
        else:
            for i, (a, b) in enumerate(zip(cur, ref)):
                if not torch.equal(a, b):
                    bad += 1
                    if bad <= 3:
                        d = (a.float() - b.float()).abs()
                        print("  MISMATCH tensor", i, "rep", rep, "max", d.max().item(), "count", int((d > 0).sum()))
    print((C, nH, H, nB, shift), "mismatching tensors over", reps - 1, "repeats:", bad, " y == y(no side outputs):", bool(torch.equal(ref[0], ref[1])))
