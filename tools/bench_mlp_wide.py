"""the wide-stage (C = 384) fused MLP branch (mlp_fused32p.hip) vs the unfused LayerNorm -> fc1 -> fc2 sequence on the stage-2 row counts of one
Swin-T W7 step (run on the MI355X): python tools/bench_mlp_wide.py [--batch 128]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esvit_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e-3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--rows", type=int, nargs="*", default=None)
    a = ap.parse_args()
    B, C, dt = a.batch, 384, torch.bfloat16
    cases = [("teacher", B * 2 * 196, False), ("student", B * (2 * 196 + 8 * 36), True)]
    if a.rows:
        cases = [("rows", r, True) for r in a.rows]
    for who, M, train in cases:
        x = torch.randn(M, C, device=dev)
        g, b = 1 + 0.1 * torch.randn(C, device=dev), 0.1 * torch.randn(C, device=dev)
        W1f, b1 = torch.randn(4 * C, C, device=dev) * 0.05, 0.1 * torch.randn(4 * C, device=dev)
        W1, W2, b2 = W1f.to(dt), (torch.randn(C, 4 * C, device=dev) * 0.05).to(dt), 0.1 * torch.randn(C, device=dev)

        def unfused(save):
            h, _, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6)
            if save:
                a1g, a1 = ops.linear_fwd(h, W1, b1, gelu=True, want_preact=True)
            else:
                a1g = ops.linear_fwd(h, W1, b1, gelu=True)
            return ops.linear_fwd(a1g, W2, b2, residual=x, out_f32=True)
        yu, yf = unfused(False), ops.mlp_fused_fwd(x, g, b, 1e-6, W1, b1, W2, b2)
        err = ((yu - yf).abs().max() / yu.abs().max()).item()
        res = dict(C=C, rows=M, who=who, rel_err=err)
        res["unfused_us"] = round(timeit(lambda: unfused(False)) * 1e6, 1)
        res["fused_us"] = round(timeit(lambda: ops.mlp_fused_fwd(x, g, b, 1e-6, W1, b1, W2, b2)) * 1e6, 1)
        res["fused_TF"] = round(16.0 * M * C * C / res["fused_us"] / 1e6, 1)
        if train:
            res["unfused_train_us"] = round(timeit(lambda: unfused(True)) * 1e6, 1)
            res["fused_train_us"] = round(timeit(lambda: ops.mlp_fused_fwd_train(x, g, b, 1e-6, W1, b1, W2, b2)) * 1e6, 1)
            res["fused_train_TF"] = round(16.0 * M * C * C / res["fused_train_us"] / 1e6, 1)
            res["fused_train_GBs"] = round(M * C * (12 + 16 + 2) / res["fused_train_us"] / 1e3)
        print(json.dumps(res), flush=True)
        del x
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
