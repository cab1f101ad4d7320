"""One GEMM shape, repeated -- for rocprofv3 PMC runs.  usage: bench_one_gemm.py KIND M N K PIPE [ITERS]
KIND: fwd | gelu | res (fp32 residual + fp32 out) | dgrad | wgrad"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esvit_amd import ops

kind, M, N, K, pipe = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 20
dev = torch.device("cuda:0")
ops.lib.esvit_debug_set_gemm_pipe(pipe)
ablate = int(os.environ.get("ABLATE", "0"))
if os.environ.get("GEMM_M256"):
    ops.lib.esvit_debug_set_gemm_m256(int(os.environ["GEMM_M256"]))
if os.environ.get("GEMM_DMA"):
    ops.debug_set_gemm_dma(int(os.environ["GEMM_DMA"]))
ops.lib.esvit_debug_set_gemm_ws_ablate(ablate)
if os.environ.get("GEMM_STAGGER"):
    ops.lib.esvit_debug_set_gemm_stagger(int(os.environ["GEMM_STAGGER"]))
if os.environ.get("GEMM_PF"):
    ops.lib.esvit_debug_set_gemm_l2_prefetch(int(os.environ["GEMM_PF"]))
if os.environ.get("GEMM_GROUP_M"):
    ops.lib.esvit_debug_set_gemm_group_m(int(os.environ["GEMM_GROUP_M"]))
bf = torch.bfloat16
x = torch.randn(M, K, device=dev).to(bf)
w = (torch.randn(N, K, device=dev) * 0.05).to(bf)
b = torch.zeros(N, device=dev)
res = torch.randn(M, N, device=dev)
dy = torch.randn(M, N, device=dev).to(bf)
fn = {"fwd": lambda: ops.linear_fwd(x, w, b),
      "gelu": lambda: ops.linear_fwd(x, w, b, gelu=True, want_preact=True),
      "res": lambda: ops.linear_fwd(x, w, b, residual=res, out_f32=True),
      "dgrad": lambda: ops.linear_dgrad(dy, w),
      "wgrad": lambda: ops.linear_wgrad(dy, x, want_bias=True)}[kind]
for _ in range(3):
    fn()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters):
    fn()
e.record()
torch.cuda.synchronize()
t = s.elapsed_time(e) / iters * 1e-3
print("%s M=%d N=%d K=%d pipe=%d ablate=%d: %.1f us  %.0f TFLOP/s" % (kind, M, N, K, pipe, ablate, t * 1e6, 2.0 * M * N * K / t / 1e12))
