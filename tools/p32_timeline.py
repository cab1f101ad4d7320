"""Where one wave of the wide fused MLP kernel (mlp_fused32p.hip) spends an iteration: loop body / waits (vmcnt, lgkmcnt) / barrier, and the kernel
time of every ablation build (tools/probe/build_p32.sh).   python tools/p32_timeline.py [--rows 50176] [--train]"""
import argparse, ctypes as C, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

dev = torch.device("cuda:0")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=50176)
    ap.add_argument("--train", action="store_true")
    a = ap.parse_args()
    M, Cc, dt = a.rows, 384, torch.bfloat16
    x = torch.randn(M, Cc, device=dev)
    g, b = 1 + 0.1 * torch.randn(Cc, device=dev), 0.1 * torch.randn(Cc, device=dev)
    W1, b1 = (torch.randn(4 * Cc, Cc, device=dev) * 0.05).to(dt), 0.1 * torch.randn(4 * Cc, device=dev)
    W2, b2 = (torch.randn(Cc, 4 * Cc, device=dev) * 0.05).to(dt), 0.1 * torch.randn(Cc, device=dev)
    y = torch.empty_like(x)
    side = [torch.empty((M, 4 * Cc), dtype=dt, device=dev), torch.empty((M, 4 * Cc), dtype=dt, device=dev), torch.empty((M, Cc), dtype=dt, device=dev),
            torch.empty(M, device=dev), torch.empty(M, device=dev)] if a.train else [None] * 5
    P = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe")
    for path in sorted(glob.glob(os.path.join(here, "libp32_probe*.so"))):
        lib = C.CDLL(path)
        lib.p32_probe_fwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p] + [C.c_void_p] * 6
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        run = lambda: lib.p32_probe_fwd(P(x), P(g), P(b), 1e-6, P(W1), P(b1), P(W2), P(b2), M, P(y), *[P(t) for t in side], st)
        for _ in range(3):
            assert run() == 0
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); run(); e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 1e3)
        tl = torch.zeros((64, 4), dtype=torch.int64, device=dev)
        assert lib.p32_probe_set_timeline(C.c_void_p(tl.data_ptr())) == 0
        assert run() == 0
        torch.cuda.synchronize()
        lib.p32_probe_set_timeline(C.c_void_p(0))
        t = tl.cpu().numpy().astype(np.float64)
        t = t[(t[:, 0] > 0) & (t[:, 3] > 0)]
        if len(t) < 4:  # a build without stamps: the kernel time is the result
            print(json.dumps({"lib": os.path.basename(path), "rows": M, "train": a.train, "kernel_us": round(sorted(ts)[len(ts) // 2], 1)}), flush=True)
            continue
        steady = t[4:-4] if len(t) > 12 else t
        ns = lambda v: float(np.mean(v))  # s_memtime ticks = shader cycles
        res = {"lib": os.path.basename(path), "rows": M, "train": a.train, "kernel_us": round(sorted(ts)[len(ts) // 2], 1), "iterations_stamped": int(len(t)),
               "iter_cyc": round(ns(steady[1:, 0] - steady[:-1, 0]), 1), "body_cyc": round(ns(steady[:, 1] - steady[:, 0]), 1), "wait_cyc": round(ns(steady[:, 2] - steady[:, 1]), 1),
               "barrier_cyc": round(ns(steady[:, 3] - steady[:, 2]), 1), "first_iter_start_to_last_end_cyc": float(t[-1, 3] - t[0, 0])}
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
