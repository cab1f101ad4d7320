"""Correctness + timing of the short-K forwards of the step through ops.linear_fwd with whatever library ESVIT_HIP_LIB names.
python tools/probe/astat_check.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import esvit_amd  # noqa: E402
from esvit_amd import ops as o  # noqa: E402

esvit_amd.set_precision("bf16")
dev = torch.device("cuda:0")
torch.manual_seed(0)


def bench(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


for (M, N, K, gelu, pre, bias) in [(87040, 1536, 384, True, True, True), (87040, 1536, 384, True, False, True), (87040, 1152, 384, False, False, True),
                                    (50176, 1152, 384, False, False, True), (21760, 2048, 256, False, False, False), (87040, 384, 384, False, False, True),
                                    (256, 65536, 256, False, False, False), (12544, 65536, 256, False, False, False)]:
    x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device=dev) * 0.1 if bias else None
    if gelu and pre:
        y, a = o.linear_fwd(x, w, b, gelu=True, want_preact=True)
    else:
        y, a = o.linear_fwd(x, w, b, gelu=gelu), None
    rows = torch.randint(0, M, (512,), device=dev)
    ref = x[rows].float() @ w.float().t()
    if b is not None:
        ref = ref + b
    err_a = (a[rows].float() - ref).abs().max().item() if a is not None else 0.0
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    err = (y[rows].float() - ref).abs().max().item()
    # the last rows / columns as well (chunk ends)
    tail = (y[-128:].float() - (torch.nn.functional.gelu(x[-128:].float() @ w.float().t() + (b if b is not None else 0)) if gelu else x[-128:].float() @ w.float().t() + (b if b is not None else 0))).abs().max().item()
    us = bench(lambda: o.linear_fwd(x, w, b, gelu=gelu, want_preact=pre) if gelu else o.linear_fwd(x, w, b))
    print("M=%6d N=%6d K=%4d gelu=%d pre=%d bias=%d: %8.1f us  %6.1f TF/s   max err %.4f (preact %.4f, last rows %.4f)" %
          (M, N, K, gelu, pre, bias, us, 2.0 * M * N * K / us / 1e6, err, err_a, tail))
    del x, w, y, a
