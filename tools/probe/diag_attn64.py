"""window_attn_fwd / bwd at head_dim 64 against oracle/ops_ref over MANY windows per head (more than one window per resident
workgroup), on memory poisoned with NaN patterns first (run on the MI355X)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from esvit_amd import ops
from oracle import ops_ref as ref
import esvit_amd.functional as Fn

dev = torch.device("cuda:0")


def poison(gb=6):
    t = [torch.full((1 << 28,), float("nan"), device=dev) for _ in range(gb)]
    del t


def rnd(shape, seed, dt=torch.float32, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev).to(dt)


def case(name, nB, H, ws, nH, hd, dt, vit_N=None):
    C = nH * hd
    if vit_N is None:
        N = ws * ws
        L = H * H
        w2t = torch.from_numpy(ops.window_maps(H, H, ws, 0)[0]).to(dev)
        nW = w2t.numel() // N
        table = rnd(((2 * ws - 1) ** 2, nH), 51) * 0.5
    else:
        N = L = vit_N
        w2t, ws, table = Fn._vit_window(N, nH, dev)
        nW = 1
    qkv = rnd((nB * L, 3 * C), 50, dt)
    qb = rnd((3 * C,), 49) * 0.5
    scale = hd ** -0.5
    poison()
    o, lse = ops.window_attn_fwd(qkv, qb, w2t, L, table, ws, None, nW, N, nH, scale)
    torch.cuda.synchronize()
    fin = bool(torch.isfinite(o.float()).all())
    bad_rows = (~torch.isfinite(o.float()).all(dim=1)).nonzero().flatten()
    # reference in chunks of images (memory)
    errs = []
    step = max(1, 256 // max(1, L * L // 4096 + 1))
    for b0 in range(0, nB, step):
        b1 = min(nB, b0 + step)
        orf = ref.window_attn_fwd(qkv[b0 * L:b1 * L].float(), qb, w2t, L, table, ws, None, nW, N, nH, scale)[0]
        d = (o[b0 * L:b1 * L].float() - orf.float())
        d = torch.where(torch.isfinite(d), d, torch.full_like(d, 1e9))
        errs.append(float(d.abs().max()))
    print("%-28s nB=%4d Bw=%5d finite=%s max_err=%.4g bad_rows=%d first_bad=%s" % (name, nB, nB * nW, fin, max(errs), bad_rows.numel(),
          bad_rows[:6].tolist()), "(images of the bad rows: %s)" % sorted(set((bad_rows // L).tolist()))[:12], flush=True)
    # backward finite-ness on the same inputs
    dout = rnd((nB * L, C), 52, dt)
    poison()
    dqkv, wsb, pad = ops.window_attn_bwd(qkv, qb, w2t, L, dout, o, lse, table, ws, None, nW, N, nH, scale)
    torch.cuda.synchronize()
    berr = []
    for b0 in range(0, nB, step):
        b1 = min(nB, b0 + step)
        sl = slice(b0 * L, b1 * L)
        orf = ref.window_attn_fwd(qkv[sl].float(), qb, w2t, L, table, ws, None, nW, N, nH, scale, want_attn=False)
        dr = ref.window_attn_bwd(qkv[sl].float(), qb, w2t, L, dout[sl].float(), orf[0], orf[1], table, ws, None, nW, N, nH, scale)[0]
        d = dqkv[sl].float() - dr.float()
        d = torch.where(torch.isfinite(d), d, torch.full_like(d, 1e9))
        berr.append(float(d.abs().max()))
    print("    bwd: dqkv finite=%s max_err=%.4g  dbias_ws finite=%s  pad finite=%s" % (bool(torch.isfinite(dqkv.float()).all()), max(berr), bool(torch.isfinite(wsb).all()),
          bool(torch.isfinite(pad).all())), flush=True)


bf = torch.bfloat16
for nB in (4, 16, 17, 32, 40):
    case("cvt s0 7x7 hd64 nH1", nB, 56, 7, 1, 64, bf)
case("cvt s1 7x7 hd64 nH3", 32, 28, 7, 3, 64, bf)
case("cvt s2 7x7 hd64 nH6", 32, 14, 7, 6, 64, bf)
case("cvt s0 fp32", 32, 56, 7, 1, 64, torch.float32)
for nB in (32, 128, 400, 1280):
    case("deit_tiny N=37 hd64 nH3", nB, 0, 0, 3, 64, bf, vit_N=37)
case("deit_small N=37 hd64 nH6", 1280, 0, 0, 6, 64, bf, vit_N=37)
case("swin hd32 7x7 nH3 (control)", 40, 56, 7, 3, 32, bf)
