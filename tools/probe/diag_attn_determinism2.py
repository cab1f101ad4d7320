import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from esvit_amd import ops
dev = torch.device("cuda:0")
dt = torch.bfloat16
for (nH, H, shift, nB, hd) in ((8, 3, 0, 16, 32), (8, 7, 0, 4, 32), (4, 6, 3, 16, 32), (4, 14, 3, 4, 32)):
    ws = 7
    N, C, L = 49, nH * hd, H * H
    w2t = torch.from_numpy(ops.window_maps(H, H, ws, shift)[0]).to(dev)
    nW = w2t.numel() // N
    reg = torch.from_numpy(ops.shift_region_ids(H, H, ws, shift)).to(dev) if shift else None
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(nB * L, 3 * C, generator=g).to(dev).to(dt)
    qb = (torch.randn(3 * C, generator=g) * 0.5).to(dev)
    table = (torch.randn(169, nH, generator=g) * 0.5).to(dev)
    dout = torch.randn(nB * L, C, generator=g).to(dev).to(dt)
    ref, bad = None, {}
    for rep in range(300):
        if rep % 7 == 0:
            junk = [torch.full((1 << 20,), float("nan"), device=dev) for _ in range(3)]
            del junk
        o, lse = ops.window_attn_fwd(qkv, qb, w2t, L, table, ws, reg, nW, N, nH, hd ** -0.5)
        dqkv, bws, pad = ops.window_attn_bwd(qkv, qb, w2t, L, dout, o, lse, table, ws, reg, nW, N, nH, hd ** -0.5)
        cur = dict(out=o.clone(), dqkv=dqkv.clone(), bws=bws.clone(), pad=pad.clone())
        if ref is None:
            ref = cur
        else:
            for k in cur:
                if not torch.equal(cur[k], ref[k]):
                    bad[k] = bad.get(k, 0) + 1
                    if bad[k] <= 2:
                        d = (cur[k].float() - ref[k].float()).abs()
                        idx = (d > 0).nonzero()
                        print("  MISMATCH", (nH, H, shift, nB), k, "rep", rep, "max", d.max().item(), "count", idx.shape[0], "first", idx[:4].tolist(), "shape", tuple(cur[k].shape))
    print((nH, H, shift, nB, hd), "mismatches over 299 repeats:", bad)
