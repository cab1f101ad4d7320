"""Times the crop producer (esvit_amd.data.DataAugmentationDINO -> esvit_aug_crops) on one MI355X at the benchmark batch:
B decoded images resident in HBM -> 2 x 224^2 + 8 x 96^2 float32 crops per image.  One JSON line: images/s of the device work
(HIP events around `iters` renderings of pre-drawn parameters), the host time of the vectorised draw, and the HBM roofline view
(algorithmic bytes = the uint8 crop boxes read + the float32 crops written).

    python tools/bench_augment.py [--batch 128] [--iters 50]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-images", type=int, default=64, help="images of the batch the Pillow baseline renders")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="the Pillow baseline repeats its sample for about this long")
    args = ap.parse_args()
    from esvit_amd import data as D, ops
    rng = np.random.default_rng(args.seed)
    B = args.batch
    # ImageNet-like decoded sizes: the short side ~ 333-500, aspect ratios around 4:3 / 3:4
    hw = [(int(h), int(w)) for h, w in zip(rng.integers(300, 520, B), rng.integers(300, 520, B))]
    images = [torch.from_numpy(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).cuda() for h, w in hw]
    packed = D.PackedImages(images)
    aug = D.DataAugmentationDINO((0.4, 1.0), (0.05, 0.4), (8,), (96,), seed=args.seed)
    t0 = time.perf_counter()
    for _ in range(5):
        draws = aug.draw(packed)
    draw_ms = (time.perf_counter() - t0) / 5 * 1e3
    dev = {S: (torch.from_numpy(rows).cuda(), mh, mw) for S, (rows, mh, mw) in draws.items()}
    planes = {S: torch.empty(p.shape[0] * (3 * S * S + 4), dtype=torch.uint8, device="cuda") for S, (p, _, _) in dev.items()}
    outs = {S: torch.empty((p.shape[0], 3, S, S), dtype=torch.float32, device="cuda") for S, (p, _, _) in dev.items()}

    def render():
        for S, (p, mh, mw) in dev.items():
            ops.aug_crops(packed.data, packed.table, p, S, mh, mw, planes=planes[S], out=outs[S])
    for _ in range(3):
        render()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        render()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    box_bytes = sum(int((rows[:, 3].astype(np.int64) * rows[:, 4] * 3).sum()) for rows, _, _ in draws.values())
    out_bytes = sum(int(o.numel() * 4) for o in outs.values())
    t0 = time.perf_counter()
    crops = aug(packed)  # end to end once more: draw + upload + render
    torch.cuda.synchronize()
    e2e_ms = (time.perf_counter() - t0) * 1e3
    # cpu_baseline: the reference's own engine -- Pillow, one crop at a time as DataAugmentationDINO.__call__ runs it in a
    # DataLoader worker (datasets/build.py:252-261) -- for the SAME draws on a bounded sample of the batch, one host thread
    cpu = None
    try:
        from oracle import augment_ref as A
        from oracle.gen_augment_golden import pil_crop
        host = [im.cpu().numpy() for im in images[:args.cpu_images]]
        jobs = []
        for S, (rows, _, _) in draws.items():
            for r in rows:
                if r[0] < len(host):
                    p = A.row_to_params(r, S)
                    if p["blur"]:  # Pillow takes a Gaussian radius: one with the same box size (the same cost; timing only)
                        p["blur_radius"] = {0: 0.7, 1: 1.8}.get(p["blur_box"][0], 2.6)
                    jobs.append((int(r[0]), p))
        t0, passes = time.perf_counter(), 0
        while host and time.perf_counter() - t0 < args.cpu_seconds:  # a bounded sample: ~10 s of Pillow work
            for src, p in jobs:
                A.to_tensor_normalize(pil_crop(host[src], p))
            passes += 1
        dt = max(time.perf_counter() - t0, 1e-9)
        cpu = None if not host else {"value": passes * len(host) / dt, "unit": "images/s", "cores": 1, "kind": "reference",
               "sample": "Pillow %s (the arithmetic the reference runs), %d images x 10 crops x %d passes, same draws, one thread, %.1f s" % (
                   __import__("PIL").__version__, len(host), passes, dt)}
    except ImportError as e:  # pragma: no cover
        cpu = {"value": None, "note": "Pillow not importable: %s" % e}
    print(json.dumps({"cpu_baseline": cpu, "metric": "images/s through the GPU crop producer (2x224^2 + 8x96^2 crops per image)", "value": B / ms * 1e3, "unit": "images/s",
                      "batch": B, "ms_per_batch": ms, "host_draw_ms": draw_ms, "end_to_end_ms": e2e_ms, "crops": len(crops),
                      "roofline": {"bound": "hbm", "achieved": (box_bytes + out_bytes) / ms / 1e6, "peak": 8000.0, "unit": "GB/s",
                                   "frac": (box_bytes + out_bytes) / ms / 1e6 / 8000.0, "box_bytes": box_bytes, "out_bytes": out_bytes}}))


if __name__ == "__main__":
    main()
