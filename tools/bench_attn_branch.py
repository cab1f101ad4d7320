"""fused attention branch (esvit_attn_branch_fwd) vs the unfused LayerNorm -> qkv -> window attention -> proj sequence on the
stage-0 / stage-1 geometries of one Swin-T W7 step (run on the MI355X):
    python tools/bench_attn_branch.py [--batch 128] [--out gpurun_out/attn_branch.jsonl]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esvit_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    B = a.batch
    ops.set_act_dtype(torch.bfloat16)
    rows = []
    # (C, nH, H, images): the student's two resolution groups and the teacher's one, stages 0 and 1
    for C, nH, H, nB, who in ((96, 3, 56, 2 * B, "224 crops"), (96, 3, 24, 8 * B, "96 crops"), (192, 6, 28, 2 * B, "224 crops"), (192, 6, 12, 8 * B, "96 crops")):
        for shift in (0, 3):
            ws, N, L = 7, 49, H * H
            w2t = torch.from_numpy(ops.window_maps(H, H, ws, shift)[0]).to(dev)
            nW = w2t.numel() // N
            reg = torch.from_numpy(ops.shift_region_ids(H, H, ws, shift)).to(dev) if shift else None
            x = torch.randn(nB * L, C, device=dev)
            g1, b1 = torch.ones(C, device=dev), torch.zeros(C, device=dev)
            Wqkv, bqkv = torch.randn(3 * C, C, device=dev) * C ** -0.5, torch.randn(3 * C, device=dev) * 0.1
            Wproj, bproj = torch.randn(C, C, device=dev) * C ** -0.5, torch.randn(C, device=dev) * 0.1
            table = torch.randn((2 * ws - 1) ** 2, nH, device=dev) * 0.5
            Wq16, Wp16 = Wqkv.to(torch.bfloat16), Wproj.to(torch.bfloat16)
            Wqp, Wpp = ops.cast_weight(Wqkv, perm32=True), ops.cast_weight(Wproj, perm32=True)
            scale = 32 ** -0.5
            frag = ops.new_bias_frag(nH, N, dev)
            y = torch.empty_like(x)

            def unfused():
                xw, _, mean, rstd = ops.layernorm_fwd(x, g1, b1, 1e-6)
                qkv = ops.linear_fwd(xw, Wq16, bqkv)
                ao, _ = ops.window_attn_fwd(qkv, bqkv, w2t, L, table, ws, reg, nW, N, nH, scale, bias_frag=frag)
                return ops.linear_fwd(ao, Wp16, bproj, residual=x, out_f32=True)

            def fused():
                return ops.attn_branch_fwd(x, g1, b1, 1e-6, Wqp, bqkv, Wpp, bproj, w2t, L, table, ws, reg, nW, N, nH, scale, bias_frag=frag, out=y)

            def fused_save():
                return ops.attn_branch_fwd(x, g1, b1, 1e-6, Wqp, bqkv, Wpp, bproj, w2t, L, table, ws, reg, nW, N, nH, scale, bias_frag=frag, out=y, save=True)

            yr = unfused()
            yf = fused()
            err = float(((yf - x) - (yr - x)).abs().max() / ((yr - x).abs().max() + 1e-12))
            r = {"C": C, "H": H, "images": nB, "shift": shift, "rows": nB * L, "windows": nB * nW, "who": who,
                 "unfused_us": round(timeit(unfused), 1), "fused_us": round(timeit(fused), 1), "fused_save_us": round(timeit(fused_save), 1), "rel_err": err}
            r["bytes_min"] = nB * L * C * 8
            r["fused_TBps"] = round(r["bytes_min"] / r["fused_us"] * 1e-6, 3)
            print(json.dumps(r), flush=True)
            rows.append(r)
            del x, y, yr, yf
    if a.out:
        with open(a.out, "a") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
