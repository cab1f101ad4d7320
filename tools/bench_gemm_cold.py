"""The GEMMs of the step with COLD operands: between two timed calls 1.5 GB of unrelated writes go through the memory system, so the
operands come from HBM as they do inside the step (tools/bench_gemm.py repeats a call and reads them from the 256 MiB Infinity
Cache).  For every case: the library's choice and the forced eight-phase loops, hot and cold.

    python tools/bench_gemm_cold.py [--kernels 0,5,6]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from esvit_amd import ops

dev = torch.device("cuda:0")
DT = torch.bfloat16


def rnd(shape, scale=1.0):
    return (torch.randn(shape, device=dev) * scale).to(DT)


def timed(fn, flush, iters=7):
    ts = []
    for _ in range(iters):
        if flush is not None:
            flush.zero_()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernels", default="0,5,6")
    args = ap.parse_args()
    kernels = [int(k) for k in args.kernels.split(",")]
    flush = torch.empty(768 * 1024 * 1024, dtype=torch.bfloat16, device=dev)
    cases = [("wgrad head fc2", "wgrad", 21760, 2048, 2048), ("wgrad last", "wgrad", 20480, 65536, 256), ("wgrad s3 fc1", "wgrad", 21760, 3072, 768),
             ("dgrad s3 fc1", "dgrad", 21760, 768, 3072), ("dgrad s2 qkv", "dgrad", 87040, 384, 1152), ("fwd s3 fc2 res", "fwd_res", 21760, 768, 3072),
             ("fwd s2 fc1 gelu", "fwd_gelu", 87040, 1536, 384), ("fwd logits", "fwd", 21760, 65536, 256)]
    for name, kind, M, N, K in cases:
        if kind == "wgrad":   # dw[N, K] = dy[M, N]^T x[M, K]
            dy, x = rnd((M, N)), rnd((M, K))
            fn = lambda: ops.linear_wgrad(dy, x, want_bias=(N != 65536))
        elif kind == "dgrad":  # dx[M, N] = dy[M, K] w[K, N]
            dy, w = rnd((M, K)), rnd((K, N), 0.05)
            fn = lambda: ops.linear_dgrad(dy, w)
        else:
            x, w, b = rnd((M, K)), rnd((N, K), 0.05), torch.randn(N, device=dev)
            if kind == "fwd_res":
                r = torch.randn((M, N), device=dev)
                fn = lambda: ops.linear_fwd(x, w, b, residual=r, out_f32=True)
            elif kind == "fwd_gelu":
                fn = lambda: ops.linear_fwd(x, w, b, gelu=True, want_preact=True)
            else:
                fn = lambda: ops.linear_fwd(x, w, None)
        d = {"case": name, "M": M, "N": N, "K": K}
        for k in kernels:
            ops.FORCE_GEMM_KERNEL = k
            try:
                fn()
                d["hot_k%d" % k] = round(timed(fn, None), 1)
                d["cold_k%d" % k] = round(timed(fn, flush), 1)
            finally:
                ops.FORCE_GEMM_KERNEL = 0
        print(json.dumps(d), flush=True)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
