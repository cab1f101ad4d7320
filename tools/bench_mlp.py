"""fused MLP forward (esvit_mlp_fused_fwd) vs the unfused LayerNorm -> fc1 -> fc2 sequence on the stage-0 / stage-1 row counts
of one Swin-T W7 step (run on the MI355X): python tools/bench_mlp.py [--batch 128]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    B = ap.parse_args().batch
    dt = torch.bfloat16
    for C, rows_s, rows_t in ((96, B * (2 * 3136 + 8 * 576), B * 2 * 3136), (192, B * (2 * 784 + 8 * 144), B * 2 * 784)):
        # ---- training branch: forward + backward, fused (recompute) vs the unfused sequence of the same step ----
        M = rows_s
        x = torch.randn(M, C, device=dev)
        gy = torch.randn(M, C, device=dev) * 0.1
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        W1f, b1 = torch.randn(4 * C, C, device=dev) * 0.05, torch.zeros(4 * C, device=dev)
        W2f, b2 = torch.randn(C, 4 * C, device=dev) * 0.05, torch.zeros(C, device=dev)
        W1, W2 = W1f.to(dt), W2f.to(dt)
        K1f = ops.mlp_fused_weight(ops.MLP_W1_FWD, W1f)
        K1, W2T, W1T = ops.mlp_fused_weight(ops.MLP_W1_BWD, W1f), ops.mlp_fused_weight(ops.MLP_W2T_BWD, W2f), ops.mlp_fused_weight(ops.MLP_W1T_BWD, W1f)
        dyb = gy.to(dt)

        def unf_fwd():
            h, _, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6)
            a1g, a1 = ops.linear_fwd(h, W1, b1, gelu=True, want_preact=True)
            return h, mean, rstd, a1g, a1, ops.linear_fwd(a1g, W2, b2, residual=x, out_f32=True)
        h, mean, rstd, a1g, a1, _ = unf_fwd()

        def unf_bwd():
            dW2, dbb = ops.linear_wgrad(dyb, a1g, want_bias=True)
            da1 = ops.linear_dgrad(dyb, W2, gelu_preact=a1)
            dW1, dbb1 = ops.linear_wgrad(da1, h, want_bias=True)
            dh = ops.linear_dgrad(da1, W1)
            return ops.layernorm_bwd_cast(dh, x, mean, rstd, g, g_in=gy)

        def fus_bwd_kernel():
            return ops.mlp_fused_bwd(x, gy, g, b, 1e-6, K1, W2T, W1T, b1)

        def fus_bwd():
            gx, gxa, xhat, a1g_, da1_ = ops.mlp_fused_bwd(x, gy, g, b, 1e-6, K1, W2T, W1T, b1)
            dW2, dbb = ops.linear_wgrad(dyb, a1g_, want_bias=True)
            G, dbb1 = ops.linear_wgrad(da1_, xhat, want_bias=True)
            return ops.ln_fold_finish(G, dbb1, W1f, g, b)
        res = dict(C=C, rows=M, who="student")
        for name, fn in (("unfused_fwd", unf_fwd), ("fused_fwd", lambda: ops.mlp_fused_fwd(x, g, b, 1e-6, K1f, b1, W2, b2)),
                         ("fused_fwd_nextnorm", lambda: ops.mlp_fused_fwd(x, g, b, 1e-6, K1f, b1, W2, b2, next_norm=(g, b))),
                         ("unfused_bwd", unf_bwd), ("fused_bwd_kernel", fus_bwd_kernel), ("fused_bwd_all", fus_bwd)):
            res[name + "_us"] = round(timeit(fn) * 1e6, 1)
        res["fwd_speedup"] = round(res["unfused_fwd_us"] / res["fused_fwd_us"], 2)
        res["bwd_speedup"] = round(res["unfused_bwd_us"] / res["fused_bwd_all_us"], 2)
        res["fused_fwd_GBs"] = round(M * C * 8 / res["fused_fwd_us"] / 1e3)
        res["fused_bwd_kernel_GBs"] = round(M * C * 32 / res["fused_bwd_kernel_us"] / 1e3)
        print(json.dumps(res), flush=True)
        del x, gy, h, a1g, a1, dyb
        torch.cuda.empty_cache()
        for M, save, who in ((rows_t, False, "teacher"),):
            x = torch.randn(M, C, device=dev)
            g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
            W1f, b1 = torch.randn(4 * C, C, device=dev) * 0.05, torch.zeros(4 * C, device=dev)
            W1, K1 = W1f.to(dt), ops.mlp_fused_weight(ops.MLP_W1_FWD, W1f)
            W2, b2 = (torch.randn(C, 4 * C, device=dev) * 0.05).to(dt), torch.zeros(C, device=dev)

            def unfused():
                h, _, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6)
                if save:
                    a1g, a1 = ops.linear_fwd(h, W1, b1, gelu=True, want_preact=True)
                else:
                    a1g = ops.linear_fwd(h, W1, b1, gelu=True)
                return ops.linear_fwd(a1g, W2, b2, residual=x, out_f32=True)
            tu = timeit(unfused)
            tf = timeit(lambda: ops.mlp_fused_fwd(x, g, b, 1e-6, K1, b1, W2, b2))
            yu, yf = unfused(), ops.mlp_fused_fwd(x, g, b, 1e-6, K1, b1, W2, b2)
            err = ((yu - yf).abs().max() / yu.abs().max()).item()
            byt = M * C * (26 if save else 8)
            print(json.dumps(dict(C=C, rows=M, who=who, unfused_us=round(tu * 1e6, 1), fused_us=round(tf * 1e6, 1), speedup=round(tu / tf, 2),
                                  fused_GBs=round(byt / tf / 1e9), fused_TF=round(16.0 * M * C * C / tf / 1e12, 1), rel_err=err)), flush=True)
            del x


if __name__ == "__main__":
    main()
