"""GEMM micro-benchmark on the shapes of one Swin-T W7 pre-training step (run on the MI355X).

    python tools/bench_gemm.py [--batch 128] [--probe] [--out FILE]

For every (layer, pass) of the step it times the library's own kernel choice and each forced main loop
(esvit_gemm_desc.kernel: 2 = 4-wave 128-row LDS-DMA, 3 = 8-wave 256-row LDS-DMA) through the same ops wrappers the
model uses (so split-K factors follow the kernel), checks that the forced loops agree with each other, and prints one
JSON line per shape.  --probe additionally times the experimental main loops of tools/probe/libgemm_probe.so
(build with tools/probe/build.sh) on plain long-K problems.  Operands are random (guide rule 25: zero-filled operands
clock higher)."""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from esvit_amd import ops

dev = torch.device("cuda:0")
DT = torch.bfloat16


def timeit(fn, iters=12, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def rnd(shape, scale=1.0):
    return (torch.randn(shape, device=dev) * scale).to(DT)


def step_shapes(B):
    """(name, pass, M, N, K, flavour) of every distinct GEMM of the student step; rows of the 2 x 224^2 + 8 x 96^2 crops"""
    out = []
    rows = [B * (2 * 3136 + 8 * 576) >> (2 * s) for s in range(4)]
    for s, C_ in enumerate((96, 192, 384, 768)):
        M = rows[s]
        for nm, N, K, fl in (("qkv", 3 * C_, C_, "bias"), ("proj", C_, C_, "res"), ("fc1", 4 * C_, C_, "gelu"), ("fc2", C_, 4 * C_, "res")):
            out.append(("s%d %s" % (s, nm), "fwd", M, N, K, fl))
            out.append(("s%d %s" % (s, nm), "dgrad", M, K, N, "gelu_bwd" if nm == "fc2" else "plain"))   # dx[M, K_in] = dy[M, N] w[N, K]
            out.append(("s%d %s" % (s, nm), "wgrad", M, N, K, "bias"))
    Mh = 170 * B
    for nm, N, K, fl in (("head fc1", 2048, 768, "gelu"), ("head fc2", 2048, 2048, "gelu"), ("head fc3", 256, 2048, "bias"), ("head last", 65536, 256, "plain")):
        out.append((nm, "fwd", Mh, N, K, fl))
        out.append((nm, "dgrad", Mh, K, N, "plain" if nm in ("head fc1", "head last") else "gelu_bwd"))
        out.append((nm, "wgrad", Mh, N, K, "bias" if nm != "head last" else "plain"))
    return out


def run_shape(name, kind, M, N, K, flavour, kernels):
    """-> dict kernel -> (seconds, output tensor sample)"""
    res = {}
    if kind == "fwd":
        x, w, b = rnd((M, K)), rnd((N, K), 0.05), torch.randn(N, device=dev)
        r = torch.randn((M, N), device=dev) if flavour == "res" else None
        if flavour == "gelu":
            fn = lambda: ops.linear_fwd(x, w, b, gelu=True, want_preact=True)[0]
        elif flavour == "res":
            fn = lambda: ops.linear_fwd(x, w, b, residual=r, out_f32=True)
        elif flavour == "bias":
            fn = lambda: ops.linear_fwd(x, w, b)
        else:
            fn = lambda: ops.linear_fwd(x, w, None)
    elif kind == "dgrad":  # dx[M, N] = dy[M, K] @ w[K, N]
        dy, w = rnd((M, K)), rnd((K, N), 0.05)
        pre = rnd((M, N)) if flavour == "gelu_bwd" else None
        fn = lambda: ops.linear_dgrad(dy, w, gelu_preact=pre)
    else:  # wgrad: dw[N, K] = dy[M, N]^T x[M, K]
        dy, x = rnd((M, N)), rnd((M, K))
        fn = lambda: ops.linear_wgrad(dy, x, want_bias=(flavour == "bias"))
    for k in kernels:
        ops.FORCE_GEMM_KERNEL = k
        try:
            o = fn()
            o = o[0] if isinstance(o, tuple) else o
            t = timeit(fn)
            res[k] = (t, o.float().flatten()[:: max(1, o.numel() // 65536)].clone())
        finally:
            ops.FORCE_GEMM_KERNEL = 0
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--probe", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default=None, help="substring filter on the shape name")
    ap.add_argument("--probe-variants", default="6,1,5,2,3,4", help="probe_gemm variants to time (the first one is the reference result)")
    ap.add_argument("--probe-cases", default=None, help="comma-separated indices into the probe case list")
    ap.add_argument("--plain", action="store_true", help="plain bf16 forward problems (random operands) through every forced main loop first")
    args = ap.parse_args()
    fh = open(args.out, "w") if args.out else None

    def emit(d):
        line = json.dumps(d)
        print(line, flush=True)
        if fh:
            fh.write(line + "\n")
            fh.flush()

    if args.plain:
        for M, N, K in ((8192, 8192, 4096), (4096, 4096, 4096), (8192, 8192, 8192), (21760, 2048, 2048), (21760, 3072, 768), (21760, 768, 3072),
                        (87040, 1536, 384), (87040, 384, 1536), (21760, 65536, 256)):
            x, w = rnd((M, K)), rnd((N, K), 0.05)
            d = {"plain": "nt", "M": M, "N": N, "K": K}
            ref = None
            for k in (2, 5, 6):
                ops.FORCE_GEMM_KERNEL = k
                try:
                    fn = lambda: ops.linear_fwd(x, w, None)
                    o = fn().float().flatten()[:: max(1, M * N // 65536)].clone()
                    t = timeit(fn)
                finally:
                    ops.FORCE_GEMM_KERNEL = 0
                ref = o if ref is None else ref
                d["us_k%d" % k] = round(t * 1e6, 1)
                d["tf_k%d" % k] = round(2.0 * M * N * K / t / 1e12, 1)
                d["err_k%d" % k] = float((o - ref).abs().max().item() / (ref.abs().max().item() + 1e-12))
            emit(d)
            del x, w
            torch.cuda.empty_cache()
    KERNELS = (0, 2, 4, 5, 6)
    tot = {k: 0.0 for k in KERNELS}
    tot["best"] = 0.0
    for name, kind, M, N, K, fl in step_shapes(args.batch):
        if args.only and args.only not in name:
            continue
        r = run_shape(name, kind, M, N, K, fl, KERNELS)
        flops = 2.0 * M * N * K
        ref = r[2][1]
        scale = ref.abs().max().item() + 1e-12
        d = {"name": name, "pass": kind, "M": M, "N": N, "K": K, "flavour": fl}
        for k, (t, o) in r.items():
            d["us_k%d" % k] = round(t * 1e6, 1)
            d["tf_k%d" % k] = round(flops / t / 1e12, 1)
            d["err_k%d" % k] = float((o - ref).abs().max().item() / scale)
        sel = ops.gemm_select(DT, M=(N if kind == "wgrad" else M), N=(K if kind == "wgrad" else N), K=(M if kind == "wgrad" else K),
                              a_kstrided=int(kind == "wgrad"), b_kstrided=int(kind != "fwd"))
        d["auto_kernel"] = sel[0]
        d["best"] = min(KERNELS[1:], key=lambda k: r[k][0])
        for k in KERNELS:
            tot[k] += r[k][0]
        tot["best"] += min(r[k][0] for k in KERNELS[1:])
        emit(d)
        torch.cuda.empty_cache()
    emit({"summary_ms_per_distinct_shape_set": {str(k): round(v * 1e3, 3) for k, v in tot.items()}})

    if args.probe:
        lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libgemm_probe.so"))
        lib.probe_gemm.restype = C.c_int
        cases = [("nt", 8192, 8192, 4096), ("nt", 21760, 768, 3072), ("nt", 21760, 3072, 768), ("nt", 87040, 1536, 384), ("nt", 87040, 384, 1536),
                 ("nn", 21760, 256, 65536), ("nn", 87040, 1536, 384), ("tn", 1536, 384, 87040), ("tn", 768, 3072, 21760), ("tn", 2048, 2048, 21760),
                 ("nt", 21760, 65536, 256), ("nt", 21760, 2048, 2048), ("nn", 21760, 768, 3072)]
        if args.probe_cases:
            cases = [cases[int(i)] for i in args.probe_cases.split(",")]
        variants = [int(v) for v in args.probe_variants.split(",")]
        for lay, M, N, K in cases:
            aks, bks = int(lay == "tn"), int(lay != "nt")
            A = rnd((K, M) if aks else (M, K))
            Bm = rnd((K, N) if bks else (N, K), 0.05)
            splitk = 0
            if lay == "tn" or K >= 16384:
                tiles = -(-M // 256) * -(-N // 256)
                splitk = max(1, 256 // tiles)
            out = torch.empty((M, N), dtype=torch.float32 if splitk > 1 else DT, device=dev)
            part = torch.empty((max(splitk, 1) * M * N,), dtype=torch.float32, device=dev) if splitk > 1 else None
            d = {"probe": lay, "M": M, "N": N, "K": K, "splitk": splitk}
            ref = None
            for v in variants:
                sk = splitk
                if v == 6 and splitk:  # 128-wide tiles: four times the tiles
                    sk = max(1, 512 // (-(-M // 128) * -(-N // 128)))
                    part6 = torch.empty((sk * M * N,), dtype=torch.float32, device=dev)
                if v == 4 and splitk:
                    sk = max(1, 256 // (-(-M // 256) * -(-N // 128)))
                    part6 = torch.empty((sk * M * N,), dtype=torch.float32, device=dev)
                desc = ops._gemm_desc(dict(A=A, B=Bm, C=out, M=M, N=N, K=K, lda=(M if aks else K), ldb=(N if bks else K), ldc=N, a_kstrided=aks,
                                           b_kstrided=bks, out_f32=int(out.dtype == torch.float32), splitk=(sk if sk > 1 else 0),
                                           partial=(part6 if v in (4, 6) and splitk else part) if sk > 1 else None))
                call = lambda: lib.probe_gemm(v, C.byref(desc), C.c_void_p(torch.cuda.current_stream().cuda_stream))
                rc = call()
                if rc != 0:
                    d["v%d" % v] = "unsupported"
                    continue
                t = timeit(call)
                o = out.float().flatten()[:: max(1, out.numel() // 65536)].clone()
                if ref is None:
                    ref = o
                d["tf_v%d" % v] = round(2.0 * M * N * K / t / 1e12, 1)
                d["us_v%d" % v] = round(t * 1e6, 1)
                d["err_v%d" % v] = float((o - ref).abs().max().item() / (ref.abs().max().item() + 1e-12))
            emit(d)
            del A, Bm, out, part
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
