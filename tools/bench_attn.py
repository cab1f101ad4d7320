"""7x7 window-attention micro-benchmark on the geometries of one Swin-T W7 pre-training step (run on the MI355X).

    python tools/bench_attn.py [--batch 128] [--out FILE]

One JSON line per (stage, crop group, shift): forward and backward time, algorithmic token traffic (fwd 8C, bwd 16C bytes
per token, bf16) and the rate it implies.  Inputs are random."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from esvit_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=2):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--out", default=None)
    ap.add_argument("--only-stage", type=int, default=None)
    ap.add_argument("--window", type=int, default=7, help="7 (Swin W7) or 14 (the W14 configurations: 224-slot kernels)")
    args = ap.parse_args()
    fh = open(args.out, "w") if args.out else None
    B, ws = args.batch, args.window
    tot_f = tot_b = 0.0
    blocks = (2, 2, 6, 2)
    for s, (C, nH) in enumerate(((96, 3), (192, 6), (384, 12), (768, 24))):
        if args.only_stage is not None and s != args.only_stage:
            continue
        for grp, (nimg, side) in (("224", (2 * B, 56 >> s)), ("96", (8 * B, 24 >> s))):
            H = W = side
            L = H * W
            for shift in (0, ws // 2):
                if shift and min(H, W) <= ws:
                    continue
                if ws == 14 and min(H, W) < 12:
                    continue  # stage 2 of the 96^2 crops (6 x 6) and stage 3 run on the 64-slot kernels
                w2t, _ = ops.window_maps(H, W, ws, shift)
                nW, N = len(w2t) // (ws * ws), ws * ws
                win2tok = torch.from_numpy(w2t).to(dev)
                region = torch.from_numpy(ops.shift_region_ids(H, W, ws, shift)).to(dev) if shift else None
                rows = nimg * L
                qkv = torch.randn((rows, 3 * C), device=dev).to(torch.bfloat16)
                qb = torch.randn(3 * C, device=dev) * 0.1
                table = torch.randn(((2 * ws - 1) ** 2, nH), device=dev) * 0.1
                dout = torch.randn((rows, C), device=dev).to(torch.bfloat16)
                scale = 32 ** -0.5
                out, lse = ops.window_attn_fwd(qkv, qb, win2tok, L, table, ws, region, nW, N, nH, scale)
                f = lambda: ops.window_attn_fwd(qkv, qb, win2tok, L, table, ws, region, nW, N, nH, scale, out=out)
                dq = torch.empty_like(qkv)
                b = lambda: ops.window_attn_bwd(qkv, qb, win2tok, L, dout, out, lse, table, ws, region, nW, N, nH, scale, dqkv_out=dq)
                tf, tb = timeit(f), timeit(b)
                d = {"stage": s, "crops": grp, "shift": shift, "rows": rows, "C": C, "nH": nH, "windows": nimg * nW,
                     "fwd_us": round(tf * 1e6, 1), "bwd_us": round(tb * 1e6, 1),
                     "fwd_GBps": round(rows * C * 8 / tf / 1e9), "bwd_GBps": round(rows * C * 16 / tb / 1e9)}
                # every block pair of a stage is (shift 0, shift ws//2); stages whose grid is one window never shift
                n_use = blocks[s] / 2 if min(H, W) > ws else blocks[s]
                tot_f += tf * n_use
                tot_b += tb * n_use
                print(json.dumps(d), flush=True)
                if fh:
                    fh.write(json.dumps(d) + "\n")
                del qkv, dout, out, dq
                torch.cuda.empty_cache()
    d = {"student_fwd_ms_per_step": round(tot_f * 1e3, 3), "student_bwd_ms_per_step": round(tot_b * 1e3, 3)}
    print(json.dumps(d))
    if fh:
        fh.write(json.dumps(d) + "\n")


if __name__ == "__main__":
    main()
