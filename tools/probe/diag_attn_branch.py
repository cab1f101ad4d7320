"""stage-by-stage comparison of the fused attention branch (side outputs on) with the unfused kernel sequence: which stage, which
slots / channels differ.  python tools/probe/diag_attn_branch.py [nH H shift nB]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from esvit_amd import ops

dev = torch.device("cuda:0")
ops.set_act_dtype(torch.bfloat16)
nH, H, shift, nB = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (3, 14, 0, 1)))
ws, hd = 7, 32
N, C, L = 49, nH * hd, H * H
torch.manual_seed(0)
w2t = torch.from_numpy(ops.window_maps(H, H, ws, shift)[0]).to(dev)
nW = w2t.numel() // N
reg = torch.from_numpy(ops.shift_region_ids(H, H, ws, shift)).to(dev) if shift else None
x = torch.randn(nB * L, C, device=dev)
g1, b1 = 1 + 0.1 * torch.randn(C, device=dev), 0.1 * torch.randn(C, device=dev)
Wqkv, bqkv = torch.randn(3 * C, C, device=dev) * C ** -0.5, torch.randn(3 * C, device=dev) * 0.5
Wproj, bproj = torch.randn(C, C, device=dev) * C ** -0.5, torch.randn(C, device=dev) * 0.5
table = torch.randn((2 * ws - 1) ** 2, nH, device=dev) * 0.5
scale = hd ** -0.5
xw, _, mean, rstd = ops.layernorm_fwd(x, g1, b1, 1e-6)
qkv = ops.linear_fwd(xw, Wqkv.to(torch.bfloat16), bqkv)
ao, _ = ops.window_attn_fwd(qkv, bqkv, w2t, L, table, ws, reg, nW, N, nH, scale)
y = ops.linear_fwd(ao, Wproj.to(torch.bfloat16), bproj, residual=x, out_f32=True)
Wqp, Wpp = ops.cast_weight(Wqkv, perm32=True), ops.cast_weight(Wproj, perm32=True)
yf, (xwf, meanf, rstdf, qkvf, aof) = ops.attn_branch_fwd(x, g1, b1, 1e-6, Wqp, bqkv, Wpp, bproj, w2t, L, table, ws, reg, nW, N, nH, scale, save=True)
yn = ops.attn_branch_fwd(x, g1, b1, 1e-6, Wqp, bqkv, Wpp, bproj, w2t, L, table, ws, reg, nW, N, nH, scale)
torch.cuda.synchronize()


def rep(name, got, ref, cols_per=32):
    got, ref = got.float(), ref.float()
    d = (got - ref).abs()
    sc = ref.abs().max().item() + 1e-12
    bad = d > 0.03 * sc
    print("%-8s max err %.3e (scale %.3e)  bad %d / %d  nan %d" % (name, d.max().item(), sc, int(bad.sum()), bad.numel(), int(torch.isnan(got).sum())))
    if bad.any() and got.dim() == 2:
        rows = bad.any(1).nonzero().flatten()
        print("   bad rows: %d of %d, first %s" % (rows.numel(), got.shape[0], rows[:12].tolist()))
        cb = bad.view(got.shape[0], -1, cols_per).any(2).any(0).nonzero().flatten().tolist() if got.shape[1] % cols_per == 0 else []
        print("   bad %d-column blocks: %s" % (cols_per, cb))
        cm = bad.any(0).nonzero().flatten()
        print("   bad columns mod %d: %s" % (cols_per, sorted(set((cm % cols_per).tolist()))))
        # slot position of the bad rows inside their window
        t2w = {}
        w2 = w2t.view(nW, N).cpu()
        for wi in range(nW):
            for sl in range(N):
                t = int(w2[wi, sl])
                if t >= 0:
                    t2w[t] = (wi, sl)
        slots = sorted(set(t2w[int(r) % L][1] for r in rows[:4000].tolist()))
        print("   window slots of bad rows: %s" % slots)
        r0 = int(rows[0])
        print("   row %d got %s" % (r0, got[r0, :8].tolist()))
        print("   row %d ref %s" % (r0, ref[r0, :8].tolist()))


rep("mean", meanf.view(-1, 1), mean.view(-1, 1), 1)
rep("rstd", rstdf.view(-1, 1), rstd.view(-1, 1), 1)
rep("xw", xwf, xw)
rep("q", qkvf[:, :C], qkv[:, :C])
rep("k", qkvf[:, C:2 * C], qkv[:, C:2 * C])
rep("v", qkvf[:, 2 * C:], qkv[:, 2 * C:])
rep("ao", aof, ao)
rep("y-x", yf - x, y - x)
rep("y-x(ns)", yn - x, y - x)
if os.environ.get("ESVIT_DIAG_DUMP"):
    torch.save({k: v.cpu() for k, v in dict(x=x, g1=g1, b1=b1, Wqkv=Wqkv, bqkv=bqkv, Wproj=Wproj, bproj=bproj, table=table, w2t=w2t, xw=xw, mean=mean, rstd=rstd, qkv=qkv,
                                              ao=ao, y=y, yf=yf, xwf=xwf, meanf=meanf, rstdf=rstdf, qkvf=qkvf, aof=aof, yn=yn).items()}, os.environ["ESVIT_DIAG_DUMP"])
