import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import esvit_amd
from esvit_amd import ops as o
esvit_amd.set_precision("bf16")
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (M, N, K, bias) in [(87040, 384, 384, True), (87040, 384, 384, False), (87040, 256, 384, False), (87040, 128, 384, False), (87040, 128, 256, False), (50176, 1152, 384, True), (87040, 768, 384, False)]:
    x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device=dev) * 0.1 if bias else None
    y = o.linear_fwd(x, w, b)
    torch.cuda.synchronize()
    bad_rows = 0
    worst = []
    for r0 in range(0, M, 8192):
        ref = x[r0:r0 + 8192].float() @ w.float().t()
        if b is not None:
            ref = ref + b
        d = (y[r0:r0 + 8192].float() - ref).abs()
        bad = (d > 0.05)
        if bad.any():
            idx = bad.nonzero()
            bad_rows += idx[:, 0].unique().numel()
            if len(worst) < 3:
                rr = idx[:, 0].unique()[:4].tolist()
                worst.append((r0, [(r0 + r, sorted(set((idx[idx[:, 0] == r][:, 1] // 16).tolist()))[:12]) for r in rr]))
    print("M=%d N=%d K=%d bias=%d: bad rows %d  e.g. %s" % (M, N, K, bias, bad_rows, worst[:2]))
