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
        for M, save, who in ((rows_t, False, "teacher"),):
            x = torch.randn(M, C, device=dev)
            g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
            W1, b1 = (torch.randn(4 * C, C, device=dev) * 0.05).to(dt), torch.zeros(4 * C, device=dev)
            W2, b2 = (torch.randn(C, 4 * C, device=dev) * 0.05).to(dt), torch.zeros(C, device=dev)

            def unfused():
                h, _, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6)
                if save:
                    a1g, a1 = ops.linear_fwd(h, W1, b1, gelu=True, want_preact=True)
                else:
                    a1g = ops.linear_fwd(h, W1, b1, gelu=True)
                return ops.linear_fwd(a1g, W2, b2, residual=x, out_f32=True)
            tu = timeit(unfused)
            tf = timeit(lambda: ops.mlp_fused_fwd(x, g, b, 1e-6, W1, b1, W2, b2))
            yu, yf = unfused(), ops.mlp_fused_fwd(x, g, b, 1e-6, W1, b1, W2, b2)
            err = ((yu - yf).abs().max() / yu.abs().max()).item()
            byt = M * C * (26 if save else 8)
            print(json.dumps(dict(C=C, rows=M, who=who, unfused_us=round(tu * 1e6, 1), fused_us=round(tf * 1e6, 1), speedup=round(tu / tf, 2),
                                  fused_GBs=round(byt / tf / 1e9), fused_TF=round(16.0 * M * C * C / tf / 1e12, 1), rel_err=err)), flush=True)
            del x


if __name__ == "__main__":
    main()
