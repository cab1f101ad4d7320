"""run-to-run determinism of the 7x7 attention kernels (forward, backward, bias / pad gradients) on padded and shifted geometries, with
poisoned allocations between runs: python tools/probe/diag_attn_determinism.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from esvit_amd import ops
dev = torch.device("cuda:0")
for dt in (torch.bfloat16, torch.float32):
    for (nH, H, shift, nB, hd) in ((1, 56, 3, 2, 32), (2, 28, 0, 2, 32), (1, 24, 3, 8, 32), (2, 12, 3, 8, 32), (4, 14, 0, 2, 32), (8, 7, 0, 2, 32), (4, 6, 3, 8, 32), (8, 3, 0, 8, 32), (3, 24, 3, 64, 32), (3, 12, 0, 16, 64)):
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
        index = torch.from_numpy(ops.relative_position_index(ws)).to(dev)
        ref = None
        bad = 0
        for rep in range(12):
            junk = [torch.full((1 << 22,), float("nan"), device=dev) for _ in range(4)]
            del junk
            o, lse = ops.window_attn_fwd(qkv, qb, w2t, L, table, ws, reg, nW, N, nH, hd ** -0.5)
            dqkv, bws, pad = ops.window_attn_bwd(qkv, qb, w2t, L, dout, o, lse, table, ws, reg, nW, N, nH, hd ** -0.5)
            padsum = pad.sum(0)
            cur = (o.clone(), dqkv.clone(), bws.sum(0).clone(), padsum.clone())
            if ref is None:
                ref = cur
            else:
                for nm, a, b in zip(("out", "dqkv", "dbias", "dpad"), cur, ref):
                    if not torch.equal(a, b):
                        bad += 1
                        d = (a.float() - b.float()).abs()
                        print("  MISMATCH", dt, (nH, H, shift, nB, hd), nm, "rep", rep, "max abs", d.max().item(), "count", int((d > 0).sum()), "nan", int(torch.isnan(a.float()).sum()))
        print(dt, (nH, H, shift, nB, hd), "mismatching tensors over 11 repeats:", bad)
