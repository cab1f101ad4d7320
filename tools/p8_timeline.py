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
    ap.add_argument("--lib", default="libp8_probe.so", help="libp8_probe.so (256 x 256), libp8n_probe.so / libp8n_probe_nostore.so (256 x 128, two sets)")
    args = ap.parse_args()
    M, N, K = args.shape
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", args.lib))
    p8n = "p8n" in args.lib
    probe_gemm = lib.p8n_probe_gemm if p8n else lib.p8_probe_gemm
    set_tl = lib.p8n_probe_set_timeline if p8n else lib.p8_probe_set_timeline
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
    ntiles = -(-M // 256) * -(-N // (128 if p8n else 256))
    tl = torch.zeros((ntiles, 8), dtype=torch.int64, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        assert probe_gemm(C.byref(desc), st) == 0
    torch.cuda.synchronize()
    assert set_tl(C.c_void_p(tl.data_ptr())) == 0
    assert probe_gemm(C.byref(desc), st) == 0
    torch.cuda.synchronize()
    set_tl(C.c_void_p(0))
    t = tl.cpu().numpy().astype(np.float64)
    us = lambda a: a * 0.01
    if p8n:
        t = t[t[:, 7] > 0]
        t0 = t[:, 0].min()
        d = t[t[:, 1] > 0]  # items that carried the previous tile's stores
        res = {"lib": args.lib, "shape": [M, N, K], "layout": args.layout, "epi": args.epi, "items": int(len(t)), "kernel_us": float(us(t[:, 7].max() - t0)),
               "item_us": float(us(t[:, 7] - t[:, 0]).mean()), "loop_us": float(us(t[:, 4] - t[:, 0]).mean()),
               "ktile0_us": float(us(d[:, 1] - d[:, 0]).mean()) if len(d) else None, "ktile1_us": float(us(d[:, 2] - d[:, 1]).mean()) if len(d) else None,
               "ktiles2to4_us_each": float(us(d[:, 3] - d[:, 2]).mean() / 3) if len(d) else None,
               "later_ktiles_us_each": float((us(d[:, 4] - d[:, 3]) / np.maximum(d[:, 5] - 5, 1)).mean()) if len(d) and K > 320 else None,
               "after_loop_us": float(us(t[:, 7] - t[:, 4]).mean())}
        t[:, 4] = t[:, 7]
    else:
        t = t[t[:, 4] > 0]
        t0 = t[:, 0].min()
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
