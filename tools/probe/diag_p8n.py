import sys, torch
sys.path.insert(0, '/root/repo')
from esvit_amd import ops
from oracle import ops_ref as ref
dev = torch.device('cuda:0')
torch.manual_seed(0)
M, N, K = 512, 256, 64
x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.1).bfloat16(); b = torch.randn(N, device=dev)
res = torch.randn(M, N, device=dev); sc = torch.rand(M // 4 + 1, device=dev)
ops.FORCE_GEMM_KERNEL = 6
for name, kw in (("res", dict(residual=res, out_f32=True)), ("res+bias", dict(residual=res, out_f32=True, bias=b)),
                 ("res+scale", dict(residual=res, out_f32=True, rowscale=sc, rows_per_sample=4))):
    bias = kw.pop("bias", None)
    got = ops.linear_fwd(x, w, bias, **kw); want = ref.linear_fwd(x, w, bias, **kw)
    d = (got - want).abs()
    print(name, "max err", d.max().item(), "bad rows", (d.max(dim=1).values > 0.05).nonzero().flatten()[:12].tolist(), "bad cols", (d.max(dim=0).values > 0.05).nonzero().flatten()[:12].tolist())
