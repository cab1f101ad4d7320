"""Times esvit_dino_ce_fwd_bwd at the region-loss shape of the benchmark step (Rs = B * 170 student rows scored against <= 2 of the
B * 98 teacher rows, out_dim 65536, bf16), image-major work order.   python tools/bench_ce.py [--batch 128] [--rows-per-image 170]"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--rows-per-image", type=int, default=170)
    ap.add_argument("--teacher-rows-per-image", type=int, default=98)
    ap.add_argument("--K", type=int, default=65536)
    args = ap.parse_args()
    from esvit_amd import ops
    B, S, Tt, K = args.batch, args.rows_per_image, args.teacher_rows_per_image, args.K
    g = torch.Generator(device="cuda").manual_seed(0)
    s = torch.randn(B * S, K, device="cuda", generator=g).bfloat16()
    t = torch.randn(B * Tt, K, device="cuda", generator=g).bfloat16()
    center = torch.zeros(K, device="cuda")
    t_max, t_lse = ops.teacher_row_stats(t, center, 1 / 0.04)
    img = torch.arange(B, device="cuda").repeat_interleave(S)
    tm = torch.stack([img * Tt + torch.randint(0, Tt // 2, (B * S,), device="cuda", generator=g),
                      img * Tt + Tt // 2 + torch.randint(0, Tt // 2, (B * S,), device="cuda", generator=g)], 1).int().contiguous()
    row_w = torch.full((B * S,), 1.0 / (B * S), device="cuda")
    order = torch.arange(B * S, device="cuda", dtype=torch.int32)

    def run():
        return ops.dino_ce(s, t, center, t_max, t_lse, tm, row_w, 10.0, 25.0, row_order=order)
    loss, ds = run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    byts = s.numel() * 2 * 2 + t.numel() * 2
    print(json.dumps({"rows": B * S, "K": K, "ms": ms, "GBps_min_traffic": byts / ms / 1e6, "loss_sum": float(loss.sum()), "ds_abs": float(ds.float().abs().mean()),
                      "variant": "lds" if os.environ.get("ESVIT_CE_LDS") else "streaming"}))


if __name__ == "__main__":
    main()
