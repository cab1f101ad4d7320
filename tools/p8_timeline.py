"""Where a persistent workgroup of the eight-phase GEMM loop (ESVIT_GEMM_P8) spends its time, per output tile: first / second k-tile,
the remaining k-tiles, the epilogue, and the gap to the next item (tools/probe/build.sh builds the stamped kernel).

    python tools/p8_timeline.py [--shape M N K] [--layout nt|nn] [--epi plain|gelu|f32]
"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from esvit_amd import ops

dev = torch.device("cuda:0")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", type=int, nargs=3, default=[87040, 1536, 384])
    ap.add_argument("--layout", default="nt")
    ap.add_argument("--epi", default="plain")
    args = ap.parse_args()
    M, N, K = args.shape
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libp8_probe.so"))
    bks = int(args.layout == "nn")
    A = (torch.randn((M, K), device=dev)).to(torch.bfloat16)
    B = (torch.randn((K, N) if bks else (N, K), device=dev) * 0.05).to(torch.bfloat16)
    f32 = args.epi == "f32"
    out = torch.empty((M, N), dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
    kw = dict(A=A, B=B, C=out, M=M, N=N, K=K, lda=K, ldb=(N if bks else K), ldc=N, b_kstrided=bks, out_f32=int(f32))
    if args.epi == "gelu":
        aux = torch.empty_like(out)
        kw.update(aux=aux, ldaux=N, epilogue=1, bias=torch.randn(N, device=dev))
    desc = ops._gemm_desc(kw)
    ntiles = -(-M // 256) * -(-N // 256)
    tl = torch.zeros((ntiles, 8), dtype=torch.int64, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        assert lib.p8_probe_gemm(C.byref(desc), st) == 0
    torch.cuda.synchronize()
    assert lib.p8_probe_set_timeline(C.c_void_p(tl.data_ptr())) == 0
    assert lib.p8_probe_gemm(C.byref(desc), st) == 0
    torch.cuda.synchronize()
    lib.p8_probe_set_timeline(C.c_void_p(0))
    t = tl.cpu().numpy().astype(np.float64)
    t = t[t[:, 4] > 0]
    t0 = t[:, 0].min()
    us = lambda a: a * 0.01
    nk = t[:, 5]
    res = {"shape": [M, N, K], "layout": args.layout, "epi": args.epi, "items": int(len(t)), "kernel_us": float(us(t[:, 4].max() - t0)),
           "ktile0_us": float(us(t[:, 1] - t[:, 0]).mean()), "ktile1_us": float(us(t[:, 2] - t[:, 1]).mean()) if K >= 128 else None,
           "other_ktiles_us_each": float((us(t[:, 3] - t[:, 2]) / np.maximum(nk - 2, 1)).mean()) if K >= 192 else None,
           "loop_us": float(us(t[:, 3] - t[:, 0]).mean()), "epilogue_us": float(us(t[:, 4] - t[:, 3]).mean())}
    # gap between the end of an item's epilogue and the start of the workgroup's next item
    gaps, firsts = [], []
    for wg in np.unique(t[:, 6]):
        rows = t[t[:, 6] == wg]
        rows = rows[np.argsort(rows[:, 0])]
        gaps.extend(us(rows[1:, 0] - rows[:-1, 4]).tolist())
        firsts.append(us(rows[0, 0] - t0))
    res["gap_to_next_item_us"] = float(np.mean(gaps)) if gaps else None
    res["first_item_start_us"] = float(np.mean(firsts))
    res["items_per_wg"] = float(len(t) / len(np.unique(t[:, 6])))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
