"""Micro-benchmark of the HIP kernels on the Swin-T / DINOHead shapes (run on the MI355X)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esvit_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    if len(sys.argv) > 2:
        ops.debug_set_gemm_dma(int(sys.argv[2]))
    if len(sys.argv) > 3:
        ops.debug_set_gemm_pipe(int(sys.argv[3]))
    only_gemm = len(sys.argv) > 4
    dt = torch.bfloat16
    res = []
    shapes = []
    for s, (C, L) in enumerate([(96, 3136), (192, 784), (384, 196), (768, 49)]):
        M = 2 * B * L
        shapes += [("s%d qkv" % s, M, 3 * C, C), ("s%d proj" % s, M, C, C), ("s%d fc1" % s, M, 4 * C, C), ("s%d fc2" % s, M, C, 4 * C)]
    shapes += [("head fc1", 170 * B, 2048, 768), ("head fc2", 170 * B, 2048, 2048), ("head fc3", 170 * B, 256, 2048),
               ("head last", 170 * B, 65536, 256)]
    if len(sys.argv) > 5:
        shapes = []
    for name, M, N, K in shapes:
        x = torch.randn(M, K, device=dev).to(dt)
        w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
        b = torch.zeros(N, device=dev)
        dy = torch.randn(M, N, device=dev).to(dt)
        fl = 2.0 * M * N * K
        t_f = timeit(lambda: ops.linear_fwd(x, w, b))
        t_d = timeit(lambda: ops.linear_dgrad(dy, w))
        t_w = timeit(lambda: ops.linear_wgrad(dy, x))
        t_t = timeit(lambda: torch.nn.functional.linear(x, w))
        byt = 2.0 * (M * K + N * K + M * N)
        res.append(dict(name=name, M=M, N=N, K=K, fwd_TF=fl / t_f / 1e12, dgrad_TF=fl / t_d / 1e12, wgrad_TF=fl / t_w / 1e12,
                        torch_TF=fl / t_t / 1e12, fwd_GBs=byt / t_f / 1e9))
        print(json.dumps(res[-1]))
        del x, w, dy
    if only_gemm and len(sys.argv) <= 5:
        return
    # attention (token-ordered; 224-crop geometries H = 56, 28, 14, 7 and the padded 96-crop ones H = 24, 12, 6, 3)
    for s, (C, nH) in enumerate([(96, 3), (192, 6), (384, 12), (768, 24)]):
        for H, nimg in ((56 >> s, 2 * B), (24 >> s, 8 * B)):
            shift = 3 if H > 7 or (H == 6) else 0
            w2t_np, _ = ops.window_maps(H, H, 7, shift)
            w2t = torch.from_numpy(w2t_np).to(dev)
            nW = w2t.numel() // 49
            L = H * H
            qkv = torch.randn(nimg * L, 3 * C, device=dev).to(dt)
            qb = torch.zeros(3 * C, device=dev)
            table = torch.randn(169, nH, device=dev)
            index = torch.from_numpy(ops.relative_position_index(7)).to(dev)
            bias = ops.relpos_bias_fwd(table, index, 49)
            mask = torch.from_numpy(ops.shift_region_ids(H, H, 7, shift)).to(dev) if shift else None
            dout = torch.randn(nimg * L, C, device=dev).to(dt)
            t_f = timeit(lambda: ops.window_attn_fwd(qkv, qb, w2t, L, table, 7, mask, nW, 49, nH, 32 ** -0.5))
            t_b = timeit(lambda: ops.window_attn_bwd(qkv, qb, w2t, L, dout, None, None, table, 7, mask, nW, 49, nH, 32 ** -0.5))
            ops.lib.esvit_debug_set_attn_bwd_waves(1)
            t_b1 = timeit(lambda: ops.window_attn_bwd(qkv, qb, w2t, L, dout, None, None, table, 7, mask, nW, 49, nH, 32 ** -0.5))
            ops.lib.esvit_debug_set_attn_bwd_waves(2)
            print(json.dumps(dict(name="attn s%d H%d" % (s, H), windows=nimg * nW, nH=nH, fwd_us=t_f * 1e6, bwd_us=t_b * 1e6, bwd_us_1wave=t_b1 * 1e6,
                                  fwd_GBs=(qkv.numel() + dout.numel()) * 2 / t_f / 1e9,
                                  bwd_GBs=(2 * qkv.numel() + dout.numel()) * 2 / t_b / 1e9)))
    # loss
    K = 65536
    s_ = torch.randn(170 * B, K, device=dev).to(dt)
    t_ = torch.randn(98 * B, K, device=dev).to(dt)
    center = torch.zeros(1, K, device=dev)
    mx, lse = ops.teacher_row_stats(t_, center, 25.0)
    tm = torch.randint(0, 98 * B, (170 * B, 2), device=dev).to(torch.int32)
    w_ = torch.full((170 * B,), 1e-3, device=dev)
    t_s = timeit(lambda: ops.teacher_row_stats(t_, center, 25.0), iters=5)
    t_c = timeit(lambda: ops.dino_ce(s_, t_, center, mx, lse, tm, w_, 10.0, 25.0), iters=5)
    print(json.dumps(dict(name="teacher_stats", ms=t_s * 1e3, GBs=t_.numel() * 2 / t_s / 1e9)))
    print(json.dumps(dict(name="dino_ce", ms=t_c * 1e3, GBs_alg=(s_.numel() * 2 * 2 + s_.numel() * 2 * 2) / t_c / 1e9)))


if __name__ == "__main__":
    main()
