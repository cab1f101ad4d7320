"""Per-workgroup phase timeline of one LDS-DMA GEMM launch (tools/probe build of the product kernel with ESVIT_PROBE_TIMELINE):
how the co-resident workgroups of a CU interleave main loop and epilogue.

    tools/probe/build.sh && python tools/gemm_timeline.py [--shape M N K] [--layout nt|nn] [--variant 6]
"""
import argparse
import collections
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
    ap.add_argument("--variant", type=int, default=6)
    ap.add_argument("--tile", type=int, nargs=2, default=[128, 128])
    ap.add_argument("--lib", default="libgemm_probe.so", help="probe build (tools/probe/build.sh ablate: libgemm_probe_skip_mfma.so, ..._skip_mfma_skip_frag.so)")
    args = ap.parse_args()
    M, N, K = args.shape
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", args.lib))
    bks = int(args.layout == "nn")
    A = (torch.randn((M, K), device=dev)).to(torch.bfloat16)
    B = (torch.randn((K, N) if bks else (N, K), device=dev) * 0.05).to(torch.bfloat16)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    desc = ops._gemm_desc(dict(A=A, B=B, C=out, M=M, N=N, K=K, lda=K, ldb=(N if bks else K), ldc=N, b_kstrided=bks))
    ntiles = -(-M // args.tile[0]) * -(-N // args.tile[1])
    tl = torch.zeros((ntiles, 8), dtype=torch.int64, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        assert lib.probe_gemm(args.variant, C.byref(desc), st) == 0
    torch.cuda.synchronize()
    assert lib.probe_set_timeline(C.c_void_p(tl.data_ptr())) == 0
    assert lib.probe_gemm(args.variant, C.byref(desc), st) == 0
    torch.cuda.synchronize()
    lib.probe_set_timeline(C.c_void_p(0))
    t = tl.cpu().numpy()
    t = t[t[:, 2] > 0]
    t0 = t[:, 0].min()
    start, mid, end = (t[:, 0] - t0) * 0.01, (t[:, 1] - t0) * 0.01, (t[:, 2] - t0) * 0.01  # us
    hw, xcc = t[:, 3], t[:, 4] & 0xf
    cu = ((xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf))
    res = {"shape": [M, N, K], "workgroups": int(len(t)), "kernel_us": float(end.max()), "cus_seen": int(len(set(cu.tolist()))),
           "main_us_mean": float((mid - start).mean()), "epi_us_mean": float((end - mid).mean()),
           "main_us_p10_p90": [float(np.percentile(mid - start, 10)), float(np.percentile(mid - start, 90))],
           "epi_us_p10_p90": [float(np.percentile(end - mid, 10)), float(np.percentile(end - mid, 90))]}
    # per CU: time with k workgroups in the main loop / in the epilogue (0.05 us grid)
    grid = np.arange(0, end.max(), 0.05)
    both = mains2 = epis2 = one_main = one_epi = idle = 0
    per_cu = collections.defaultdict(list)
    for i in range(len(t)):
        per_cu[int(cu[i])].append(i)
    for c, idx in per_cu.items():
        nm = np.zeros_like(grid)
        ne = np.zeros_like(grid)
        for i in idx:
            nm += (grid >= start[i]) & (grid < mid[i])
            ne += (grid >= mid[i]) & (grid < end[i])
        both += ((nm >= 1) & (ne >= 1)).sum()
        mains2 += ((nm >= 2) & (ne == 0)).sum()
        epis2 += ((ne >= 2) & (nm == 0)).sum()
        one_main += ((nm == 1) & (ne == 0)).sum()
        one_epi += ((ne == 1) & (nm == 0)).sum()
        idle += ((nm == 0) & (ne == 0)).sum()
    tot = float(both + mains2 + epis2 + one_main + one_epi + idle)
    res["cu_time_fractions"] = {"main+epilogue overlapped": both / tot, "two in main loop": mains2 / tot, "two in epilogue": epis2 / tot,
                                "one in main loop only": one_main / tot, "one in epilogue only": one_epi / tot, "idle": idle / tot}
    res["wgs_per_cu_mean"] = float(np.mean([len(v) for v in per_cu.values()]))
    # turnaround: with two workgroups resident per CU, the time between the end of a workgroup and the start stamp of the workgroup that
    # takes its place (the k-th start on a CU, k >= 2, follows the (k-2)-th end)
    gaps = []
    for c, idx in per_cu.items():
        s_sorted = np.sort(start[idx])
        e_sorted = np.sort(end[idx])
        for k in range(2, len(idx)):
            gaps.append(s_sorted[k] - e_sorted[k - 2])
    if gaps:
        res["turnaround_us_mean_p10_p90"] = [float(np.mean(gaps)), float(np.percentile(gaps, 10)), float(np.percentile(gaps, 90))]
    res["wg_lifetime_us_mean"] = float((end - start).mean())
    # first CU's first eight workgroups, for eyeballing
    c0 = sorted(per_cu)[0]
    res["example_cu"] = [[round(float(start[i]), 2), round(float(mid[i]), 2), round(float(end[i]), 2)] for i in sorted(per_cu[c0], key=lambda i: start[i])[:10]]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
