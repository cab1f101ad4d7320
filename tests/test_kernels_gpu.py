"""-m gpu: every HIP kernel behind the C-ABI vs the plain-PyTorch restatement (oracle/ops_ref.py)
on identical seeded inputs.  fp32 mode must agree to fp32 round-off; bf16 mode to bf16 round-off
(tolerances written per test)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _tol(dt, f32=2e-5, bf=1.5e-2):
    return f32 if dt == torch.float32 else bf


def _close(name, got, ref, tol):
    got, ref = got.float(), ref.float()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item()
    assert math.isfinite(err), "%s: non-finite output" % name
    assert err <= tol * scale, "%s: max err %.3e vs scale %.3e (rel %.3e > tol %.1e)" % (name, err, scale, err / scale, tol)


@pytest.fixture(scope="module")
def mods(lib_built):
    from esvit_amd import ops
    from oracle import ops_ref
    return ops, ops_ref


def _rand(shape, dev, seed, dt=torch.float32, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev).to(dt)


@pytest.fixture(params=[0, 3, 2, 4, 5], ids=["auto", "dma8", "dma4", "dma4w", "p8"])
def gemm_path(request, mods):
    """every bf16 GEMM main loop on every shape: the library's own choice, the 8-wave 256-row LDS-DMA loop (forward / dgrad
    layouts; it is not instantiated for the weight gradients, which then take the library's choice) and the 4-wave
    128-row LDS-DMA loop in both wave layouts (2 x 2; whole-width wave rows for N % 96 == 0) and the 256 x 256 eight-phase loop
    (where K is a whole number of 64-deep k-tiles and the call has no row map / fused bias gradient; else the library's choice), forced through
    esvit_gemm_desc.kernel (the library keeps no state); the exact-fp32 mode has one main loop (the register-staged one)"""
    ops, _ = mods
    ops.FORCE_GEMM_KERNEL = request.param
    yield request.param
    ops.FORCE_GEMM_KERNEL = 0


def _skip_redundant(gemm_path, dt):
    if dt == torch.float32 and gemm_path != 0:
        pytest.skip("the fp32 mode has one main loop")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,N,K", [(300, 96, 96), (257, 288, 96), (128, 384, 192), (1000, 256, 2048), (64, 64, 48), (520, 1024, 256),
                                   (777, 2048, 768), (130, 96, 384), (33000, 384, 200), (37000, 256, 128)])
def test_gemm_nt(mods, gemm_path, dt, M, N, K):
    _skip_redundant(gemm_path, dt)
    ops, ref = mods
    dev = _dev()
    x, w, b = _rand((M, K), dev, 1, dt), _rand((N, K), dev, 2, dt, 0.1), _rand((N,), dev, 3)
    _close("nt", ops.linear_fwd(x, w, b), ref.linear_fwd(x, w, b), _tol(dt))
    y, pre = ops.linear_fwd(x, w, b, gelu=True, want_preact=True)
    yr, prer = ref.linear_fwd(x, w, b, gelu=True, want_preact=True)
    _close("nt+gelu", y, yr, _tol(dt))
    _close("nt preact", pre, prer, _tol(dt))
    y, pre = ops.linear_fwd(x, w, b, gelu=True, want_preact=True, quick=True)
    yr, prer = ref.linear_fwd(x, w, b, gelu=True, want_preact=True, quick=True)
    _close("nt+quickgelu", y, yr, _tol(dt))
    _close("nt quick preact", pre, prer, _tol(dt))
    res = _rand((M, N), dev, 4)
    _close("nt+res f32", ops.linear_fwd(x, w, b, residual=res, out_f32=True), ref.linear_fwd(x, w, b, residual=res, out_f32=True), _tol(dt, bf=5e-3))


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 256, 128), (1024, 768, 192), (704, 520, 320), (2816, 1536, 384), (5000, 2304, 768),
                                   (3000, 2048, 2048), (264, 65536 // 8, 256)])
def test_gemm_p8(mods, M, N, K):
    """the 256 x 256 eight-phase loop (ESVIT_GEMM_P8) on its own: one, two, odd and many k-tiles (the prologue requests seven half-tiles
    ahead, the last two k-tiles request nothing), interior and ragged tiles, every epilogue kind, all three operand layouts, split-K"""
    ops, ref = mods
    dev = _dev()
    dt = torch.bfloat16
    ops.FORCE_GEMM_KERNEL = 5
    try:
        assert ops.gemm_select(dt, M=M, N=N, K=K)[0] == 5
        x, w, b = _rand((M, K), dev, 1, dt), _rand((N, K), dev, 2, dt, 0.1), _rand((N,), dev, 3)
        _close("p8 nt", ops.linear_fwd(x, w, b), ref.linear_fwd(x, w, b), _tol(dt))
        _close("p8 nt no bias", ops.linear_fwd(x, w, None), ref.linear_fwd(x, w, None), _tol(dt))
        y, pre = ops.linear_fwd(x, w, b, gelu=True, want_preact=True)
        yr, prer = ref.linear_fwd(x, w, b, gelu=True, want_preact=True)
        _close("p8 nt+gelu", y, yr, _tol(dt))
        _close("p8 preact", pre, prer, _tol(dt))
        res = _rand((M, N), dev, 4)
        sc = torch.rand(M // 4 + 1, device=dev)
        kw = dict(residual=res, rowscale=sc, rows_per_sample=4, out_f32=True)
        _close("p8 nt+res f32", ops.linear_fwd(x, w, b, **kw), ref.linear_fwd(x, w, b, **kw), _tol(dt, bf=5e-3))
        dy, wd = _rand((M, K), dev, 5, dt), _rand((K, N), dev, 6, dt, 0.1)
        _close("p8 dgrad", ops.linear_dgrad(dy, wd), ref.linear_dgrad(dy, wd), _tol(dt))
        _close("p8 dgrad f32", ops.linear_dgrad(dy, wd, out_f32=True), ref.linear_dgrad(dy, wd, out_f32=True), _tol(dt))
        pre = _rand((M, N), dev, 7, dt)
        _close("p8 dgrad+gelu'", ops.linear_dgrad(dy, wd, gelu_preact=pre), ref.linear_dgrad(dy, wd, gelu_preact=pre), _tol(dt))
        rows = K * 8  # weight gradient [M', N'] = dy^T x over `rows` (split-K inside linear_wgrad)
        Mw, Nw = min(M, 1024), min(N, 1024)
        dyw, xw = _rand((rows, Mw), dev, 8, dt), _rand((rows, Nw), dev, 9, dt)
        assert ops.gemm_select(dt, M=Mw, N=Nw, K=rows, a_kstrided=1, b_kstrided=1)[0] == 5
        _close("p8 wgrad", ops.linear_wgrad(dyw, xw), ref.linear_wgrad(dyw, xw), _tol(dt, bf=2e-3))
        dw, db = ops.linear_wgrad(dyw, xw, want_bias=True)
        dwr, dbr = ref.linear_wgrad(dyw, xw, want_bias=True)
        _close("p8 wgrad(+bias) dw", dw, dwr, _tol(dt, bf=2e-3))
        _close("p8 wgrad(+bias) db", db, dbr, _tol(dt, bf=2e-3))
        acc = _rand((Mw, Nw), dev, 10)
        acc_ref = acc.clone()
        _close("p8 wgrad acc", ops.linear_wgrad(dyw, xw, out=acc, accumulate=True), ref.linear_wgrad(dyw, xw, out=acc_ref, accumulate=True),
               _tol(dt, bf=2e-3))
    finally:
        ops.FORCE_GEMM_KERNEL = 0


def test_gemm_grouped_tile_order(mods, gemm_path):
    """tile_coords(): the grouped tile order (12..64 column tiles) is a permutation of the tiles -- same result, including
    ragged last groups (row blocks % 16 != 0) and ragged edge tiles; also long-K shapes with few tiles (the 8-wave loop's
    ragged k-tiles) and a split-K dgrad over a 65536-long reduction"""
    ops, ref = mods
    dev = _dev()
    dt = torch.bfloat16
    for M, N, K in ((1500, 2304, 128), (700, 4096, 64), (2700, 2048 + 96, 192), (2700, 3072, 776), (5000, 1536, 1000)):
        x, w, b = _rand((M, K), dev, 1, dt), _rand((N, K), dev, 2, dt, 0.1), _rand((N,), dev, 3)
        _close("grouped nt", ops.linear_fwd(x, w, b), ref.linear_fwd(x, w, b), _tol(dt))
        y, pre = ops.linear_fwd(x, w, b, gelu=True, want_preact=True)
        yr, prer = ref.linear_fwd(x, w, b, gelu=True, want_preact=True)
        _close("grouped nt+gelu", y, yr, _tol(dt))
        _close("grouped preact", pre, prer, _tol(dt))
        dy, wd = _rand((M, K), dev, 5, dt), _rand((K, N), dev, 6, dt, 0.1)
        _close("grouped dgrad", ops.linear_dgrad(dy, wd), ref.linear_dgrad(dy, wd), _tol(dt))
    dy, wd = _rand((1300, 8192), dev, 5, dt), _rand((8192, 256), dev, 6, dt, 0.05)
    _close("long-K dgrad (split-K)", ops.linear_dgrad(dy, wd), ref.linear_dgrad(dy, wd), _tol(dt))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,N,K", [(300, 96, 384), (257, 192, 576), (1000, 2048, 256), (130, 768, 3072), (40000, 192, 96), (3000, 384, 1536)])
def test_gemm_dgrad(mods, gemm_path, dt, M, N, K):
    """dx[M,N] = dy[M,K] @ w[K,N] (B read k-strided with ds_read_b64_tr_b16)."""
    _skip_redundant(gemm_path, dt)
    ops, ref = mods
    dev = _dev()
    dy, w = _rand((M, K), dev, 5, dt), _rand((K, N), dev, 6, dt, 0.1)
    _close("dgrad", ops.linear_dgrad(dy, w), ref.linear_dgrad(dy, w), _tol(dt))
    pre = _rand((M, N), dev, 7, dt)
    _close("dgrad+gelu'", ops.linear_dgrad(dy, w, gelu_preact=pre), ref.linear_dgrad(dy, w, gelu_preact=pre), _tol(dt))
    _close("dgrad+quickgelu'", ops.linear_dgrad(dy, w, gelu_preact=pre, quick=True), ref.linear_dgrad(dy, w, gelu_preact=pre, quick=True),
           _tol(dt))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("rows,Nout,Kin", [(392, 288, 96), (3136, 96, 384), (1000, 256, 2048), (98, 768, 768), (6272, 64, 48),
                                           (25088, 1152, 384), (12545 * 8, 192, 576), (20000, 384, 1536), (9000, 2048, 768),
                                           (4100, 768, 2304), (7000, 1000, 264)])
def test_gemm_wgrad(mods, gemm_path, dt, rows, Nout, Kin):
    """dw = dy^T x, both operands k-strided (transpose reads), split-K over the rows; the shapes cover every tile of the
    8-wave loop (192 / 256 rows x 192 / 256 columns, ragged edges) and of the 4-wave loop"""
    _skip_redundant(gemm_path, dt)
    ops, ref = mods
    dev = _dev()
    dy, x = _rand((rows, Nout), dev, 8, dt), _rand((rows, Kin), dev, 9, dt)
    _close("wgrad", ops.linear_wgrad(dy, x), ref.linear_wgrad(dy, x), _tol(dt, f32=5e-5, bf=2e-3))
    dw, db = ops.linear_wgrad(dy, x, want_bias=True)
    dwr, dbr = ref.linear_wgrad(dy, x, want_bias=True)
    _close("wgrad(+bias) dw", dw, dwr, _tol(dt, f32=5e-5, bf=2e-3))
    _close("wgrad(+bias) db", db, dbr, _tol(dt, f32=5e-5, bf=2e-3))
    acc = _rand((Nout, Kin), dev, 10)
    acc_ref = acc.clone()
    _close("wgrad acc", ops.linear_wgrad(dy, x, out=acc, accumulate=True), ref.linear_wgrad(dy, x, out=acc_ref, accumulate=True),
           _tol(dt, f32=5e-5, bf=2e-3))


@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_rowmap_scatter(mods, gemm_path, dt):
    """window rows -> token rows with pad rows dropped, DropPath scale and residual (proj epilogue)."""
    _skip_redundant(gemm_path, dt)
    ops, ref = mods
    dev = _dev()
    H, ws, shift, C, nB = 6, 7, 3, 96, 3
    win2tok, _ = ops.window_maps(H, H, ws, shift)
    rm = torch.from_numpy(win2tok).to(dev)
    M = nB * rm.numel()
    x, w, b = _rand((M, C), dev, 11, dt), _rand((C, C), dev, 12, dt, 0.1), _rand((C,), dev, 13)
    res = _rand((nB * H * H, C), dev, 14)
    sc = torch.tensor([1.0, 0.0, 1.0 / 0.9], device=dev)
    kw = dict(residual=res, rowmap=rm, rowmap_tokens=H * H, out_rows=nB * H * H, rowscale=sc, rows_per_sample=H * H, out_f32=True)
    _close("scatter", ops.linear_fwd(x, w, b, **kw), ref.linear_fwd(x, w, b, **kw), _tol(dt, bf=5e-3))


def test_batched_sim(mods):
    ops, ref = mods
    dev = _dev()
    a, b = _rand((6, 170, 768), dev, 15), _rand((6, 49, 768), dev, 16)
    got, want = ops.batched_nt(a, b, 64), ref.batched_nt(a, b, 64)
    _close("sim", got[:, :, :49], want[:, :, :49], 2e-5)


@pytest.mark.parametrize("C,M", [(96, 128 * 37), (192, 128 * 21), (96, 1000), (192, 77), (96, 128 * 400 + 33), (128, 128 * 33 + 5), (256, 128 * 17), (256, 61)])
def test_mlp_fused_fwd(mods, C, M):
    """fused LayerNorm -> fc1 + GELU -> fc2 + residual (esvit_mlp_fused_fwd, the teacher's narrow stages) vs the unfused op
    sequence it replaces; full and ragged row tiles, with and without per-row scales"""
    ops, ref = mods
    dev = _dev()
    dt = torch.bfloat16
    assert ops.mlp_fused_supported(dt, C) and not ops.mlp_fused_supported(dt, 768) and not ops.mlp_fused_supported(torch.float32, C)
    assert ops.mlp_fused_supported(dt, C, backward=True) and not ops.mlp_fused_supported(dt, 384, backward=True)
    x = _rand((M, C), dev, 60) * 1.5 + 0.3
    g, b = 1.0 + 0.2 * _rand((C,), dev, 61), 0.1 * _rand((C,), dev, 62)
    W1f, b1 = _rand((4 * C, C), dev, 63, torch.float32, 0.08), 0.1 * _rand((4 * C,), dev, 64)
    W2, b2 = _rand((C, 4 * C), dev, 65, dt, 0.05), 0.1 * _rand((C,), dev, 66)
    W1, W1k = W1f.to(dt), ops.mlp_fused_weight(ops.MLP_W1_FWD, W1f)  # natural cast (restatement) / the kernels' own format
    for rs in (None, (torch.rand(M, generator=torch.Generator().manual_seed(67)) > 0.3).float().div(0.7).to(dev)):
        got = ops.mlp_fused_fwd(x, g, b, 1e-6, W1k, b1, W2, b2, rowscale=rs)
        want = ref.mlp_fused_fwd(x, g, b, 1e-6, W1, b1, W2, b2, rowscale=rs)
        _close("mlp y", got, want, 4e-3)  # fp32 output; the hidden activation is rounded to bf16 on both sides
        # second output: LayerNorm(y) with the next block's norm1 parameters + its row statistics
        gn, bn = 1.0 + 0.3 * _rand((C,), dev, 68), 0.2 * _rand((C,), dev, 69)
        got2, (xw, mean, rstd) = ops.mlp_fused_fwd(x, g, b, 1e-6, W1k, b1, W2, b2, rowscale=rs, next_norm=(gn, bn))
        assert torch.equal(got2, got)
        xw_r, _, mean_r, rstd_r = ref.layernorm_fwd(got, gn, bn, 1e-6, dtype=dt)  # (statistics of the kernel's own y)
        _close("next-norm xw", xw, xw_r, 8e-3)
        _close("next-norm mean", mean, mean_r, 2e-5)
        _close("next-norm rstd", rstd, rstd_r, 2e-5)


@pytest.mark.parametrize("M", [128 * 5, 77, 128 * 392, 128 * 256 + 1, 31])
def test_mlp_fused_fwd_wide(mods, M):
    """the wide-stage (C = 384) fused MLP branch (mlp_fused32p.hip: one wave per SIMD, the two products and the GELU between them software-
    pipelined): inference pass (esvit_mlp_fused_fwd) and training pass (esvit_mlp_fused_fwd_train: y + the operands of the unfused backward)
    vs the unfused op sequence; full and ragged row tiles, one round and several rounds of workgroups, with and without per-row scales"""
    ops, ref = mods
    dev = _dev()
    dt = torch.bfloat16
    C = 384
    assert ops.mlp_fused_supported(dt, C) and ops.mlp_fused_train_supported(dt, C) and not ops.mlp_fused_train_supported(dt, 192)
    x = _rand((M, C), dev, 160) * 1.5 + 0.3
    g, b = 1.0 + 0.2 * _rand((C,), dev, 161), 0.1 * _rand((C,), dev, 162)
    W1f, b1 = _rand((4 * C, C), dev, 163, torch.float32, 0.06), 0.1 * _rand((4 * C,), dev, 164)
    W2, b2 = _rand((C, 4 * C), dev, 165, dt, 0.04), 0.1 * _rand((C,), dev, 166)
    W1, W1k = W1f.to(dt), ops.mlp_fused_weight(ops.MLP_W1_FWD, W1f)
    assert torch.equal(W1, W1k)  # the plain cast at this width
    for rs in (None, (torch.rand(M, generator=torch.Generator().manual_seed(167)) > 0.3).float().div(0.7).to(dev)):
        want = ref.mlp_fused_fwd_train(x, g, b, 1e-6, W1, b1, W2, b2, rowscale=rs)
        got = ops.mlp_fused_fwd(x, g, b, 1e-6, W1k, b1, W2, b2, rowscale=rs)
        assert bool(torch.isfinite(got).all())
        _close("wide mlp y", got, want[0], 4e-3)  # fp32 output; the hidden activation is rounded to bf16 on both sides
        tr = ops.mlp_fused_fwd_train(x, g, b, 1e-6, W1k, b1, W2, b2, rowscale=rs)
        # the two instantiations run the same arithmetic; hipcc contracts the LayerNorm statistics differently in them, so a few rows in ten
        # thousand get a neighbouring bf16 rounding of some LN(x) values (observed: 8 of 50176 rows, max |dy| 6e-3)
        same_rows = (tr[0] == got).all(dim=1).float().mean().item()
        assert same_rows > 0.999, "the side outputs change y (%.4f of the rows equal)" % same_rows
        _close("wide mlp y, training pass", tr[0], got, 2e-3)
        for name, a, r, tol in zip(("a1", "a1g", "h", "mean", "rstd"), tr[1:], want[1:], (8e-3, 8e-3, 8e-3, 2e-5, 2e-5)):
            _close("wide mlp " + name, a, r, tol)
    # extreme pre-activations: the rational CDF of the kernel's GELU is clamped at |a| = 4.5 (its tail beyond: 3.4e-6)
    xe = torch.zeros((128, C), device=dev)
    xe[:, 0] = torch.linspace(-40, 40, 128, device=dev)
    W1e = torch.zeros((4 * C, C), device=dev)
    W1e[:, 0] = torch.linspace(-1, 1, 4 * C, device=dev)
    ones, zeros = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    tre = ops.mlp_fused_fwd_train(xe, ones, zeros, 1e-6, W1e.to(dt), b1 * 0, W2, b2)
    wante = ref.mlp_fused_fwd_train(xe, ones, zeros, 1e-6, W1e.to(dt), b1 * 0, W2, b2)
    _close("wide mlp a1g, extreme", tre[2], wante[2], 8e-3)
    _close("wide mlp y, extreme", tre[0], wante[0], 4e-3)


def test_mlp_fused_fwd_wide_is_bit_reproducible(mods):
    """the wide fused MLP at the step's occupancy (87040 rows = 2.66 rounds of 256 workgroups), 15 launches of each entry point on the same inputs:
    identical to the bit.  Its MFMAs are asm statements whose results compiler-generated code reads (accumulator reads, the bias operand) -- the two
    orderings that were wrong while the kernel was written (a1 off by O(1) in the last two chunks; stale bias operand) showed up as run-to-run
    differences at exactly this size, not at 640 rows."""
    ops, _ = mods
    dev = _dev()
    dt, C, M = torch.bfloat16, 384, 87040
    x = _rand((M, C), dev, 170) * 1.5 + 0.3
    g, b = 1.0 + 0.2 * _rand((C,), dev, 171), 0.1 * _rand((C,), dev, 172)
    W1, b1 = _rand((4 * C, C), dev, 173, dt, 0.06), 0.1 * _rand((4 * C,), dev, 174)
    W2, b2 = _rand((C, 4 * C), dev, 175, dt, 0.04), 0.1 * _rand((C,), dev, 176)
    first = None
    for rep in range(15):
        y = ops.mlp_fused_fwd(x, g, b, 1e-6, W1, b1, W2, b2)
        tr = ops.mlp_fused_fwd_train(x, g, b, 1e-6, W1, b1, W2, b2)
        cur = [y] + list(tr)
        if first is None:
            first = [t.clone() for t in cur]
            assert all(bool(torch.isfinite(t.float()).all()) for t in cur)
        else:
            for i, (a, r) in enumerate(zip(cur, first)):
                assert torch.equal(a, r), "tensor %d differs between launch 0 and launch %d" % (i, rep)


@pytest.mark.parametrize("C,M", [(96, 128 * 29), (192, 128 * 9), (96, 1000), (192, 77), (96, 128 * 300 + 33), (128, 128 * 19 + 7), (256, 128 * 11), (256, 45)])
def test_mlp_fused_bwd(mods, C, M):
    """data-gradient path of the fused MLP branch (esvit_mlp_fused_bwd) vs its torch restatement: dL/dx, its activation-dtype
    copy, xhat, GELU(A), dA; full and ragged row tiles, with and without DropPath row factors.  Then the whole branch backward
    (two weight-gradient GEMMs + esvit_ln_fold_finish) against torch autograd of the unfused fp32 formula"""
    ops, ref = mods
    dev = _dev()
    dt = torch.bfloat16
    x = _rand((M, C), dev, 70) * 1.5 + 0.3
    gy = _rand((M, C), dev, 71) * 0.5
    g, b = 1.0 + 0.2 * _rand((C,), dev, 72), 0.1 * _rand((C,), dev, 73)
    W1f, b1 = _rand((4 * C, C), dev, 74, torch.float32, 0.08), 0.1 * _rand((4 * C,), dev, 75)
    W2f = _rand((C, 4 * C), dev, 76, torch.float32, 0.05)
    W1, W2 = W1f.to(dt), W2f.to(dt)
    W1T, W2T = W1.t().contiguous(), W2.t().contiguous()
    # esvit_cast_weight: transpose and / or the 32-block channel order of the 16-token kernels
    perm = torch.tensor([32 * (p >> 5) + (4 * ((p >> 3) & 3) + (p & 7) if (p & 7) < 4 else 16 + 4 * ((p >> 3) & 3) + (p & 7) - 4) for p in range(C)], device=dev)
    assert torch.equal(ops.cast_weight(W1f), W1) and torch.equal(ops.cast_weight(W1f, transpose=True), W1T)
    assert torch.equal(ops.cast_weight(W1f, perm32=True), W1[:, perm]) and torch.equal(ops.cast_weight(W2f, transpose=True, perm32=True), W2T[:, perm])
    K1, K2T, K1T = (ops.mlp_fused_weight(k, w) for k, w in ((ops.MLP_W1_BWD, W1f), (ops.MLP_W2T_BWD, W2f), (ops.MLP_W1T_BWD, W1f)))  # the kernels' own format
    assert torch.equal(K1, W1[:, perm]) and torch.equal(K2T, W2T[:, perm]) and torch.equal(K1T, W1T)
    gen = torch.Generator().manual_seed(77)
    for rs_mlp, rs_out in ((None, None), ((torch.rand(M, generator=gen) > 0.3).float().div(0.7).to(dev), (torch.rand(M, generator=gen) > 0.2).float().div(0.8).to(dev))):
        got = ops.mlp_fused_bwd(x, gy, g, b, 1e-6, K1, K2T, K1T, b1, rowscale_mlp=rs_mlp, rowscale_out=rs_out)
        want = ref.mlp_fused_bwd(x, gy, g, b, 1e-6, W1, W2T, W1T, b1, rowscale_mlp=rs_mlp, rowscale_out=rs_out)
        for name, a, r, tol in zip(("gx", "gx_act", "xhat", "a1g", "da1"), got, want, (6e-3, 1e-2, 8e-3, 8e-3, 1.2e-2)):
            _close("mlp bwd " + name, a, r, tol)
    # the whole branch against autograd (fp32 formula on the same bf16-rounded weights)
    gx, gxa, xhat, a1g, da1 = ops.mlp_fused_bwd(x, gy, g, b, 1e-6, K1, K2T, K1T, b1)
    dyb = gy.to(dt)
    dW2, db2 = ops.linear_wgrad(dyb, a1g, want_bias=True)
    G, db1 = ops.linear_wgrad(da1, xhat, want_bias=True)
    dW1, dg, dbeta = ops.ln_fold_finish(G, db1, W1.float(), g, b)
    xa = x.clone().requires_grad_(True)
    prm = [t.clone().float().requires_grad_(True) for t in (g, b, W1, b1, W2)]
    h = torch.nn.functional.layer_norm(xa, (C,), prm[0], prm[1], 1e-6)
    y = xa + torch.nn.functional.gelu(h @ prm[2].t() + prm[3]) @ prm[4].t()
    y.backward(gy)
    _close("branch dx", gx, xa.grad, 1.5e-2)
    for name, a, r in (("dgamma", dg, prm[0].grad), ("dbeta", dbeta, prm[1].grad), ("dW1", dW1, prm[2].grad), ("db1", db1, prm[3].grad),
                       ("dW2", dW2, prm[4].grad), ("db2", db2, gy.sum(0))):
        _close("branch " + name, a, r, 2e-2)


def test_ln_fold_finish(mods):
    ops, ref = mods
    dev = _dev()
    J, C = 384, 96
    G, db, W = _rand((J, C), dev, 80), _rand((J,), dev, 81), _rand((J, C), dev, 82)
    g, b = 1.0 + 0.2 * _rand((C,), dev, 83), 0.1 * _rand((C,), dev, 84)
    want = ref.ln_fold_finish(G.clone(), db, W, g, b)
    got = ops.ln_fold_finish(G.clone(), db, W, g, b)
    for name, a, r in zip(("dW", "dgamma", "dbeta"), got, want):
        _close("ln fold " + name, a, r, 2e-5)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("C", [96, 192, 384, 768, 1536])
def test_layernorm(mods, dt, C):
    ops, ref = mods
    dev = _dev()
    rows = 333
    x = _rand((rows, C), dev, 20) * 2 + 0.5
    g, b = _rand((C,), dev, 21) * 0.2 + 1, _rand((C,), dev, 22) * 0.1
    y, yf, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6, want_f32=True, dtype=dt)
    yr, yfr, meanr, rstdr = ref.layernorm_fwd(x, g, b, 1e-6, want_f32=True, dtype=dt)
    _close("ln y", y, yr, _tol(dt, bf=8e-3))
    _close("ln yf", yf, yfr, 2e-5)
    _close("ln mean", mean, meanr, 2e-5)
    _close("ln rstd", rstd, rstdr, 2e-5)
    dy = _rand((rows, C), dev, 23, dt)
    gin = _rand((rows, C), dev, 24)
    dx, dg, db = ops.layernorm_bwd(dy, x, meanr, rstdr, g, g_in=gin)
    dxr, dgr, dbr = ref.layernorm_bwd(dy, x, meanr, rstdr, g, g_in=gin)
    _close("ln dx", dx, dxr, 5e-5)
    _close("ln dgamma", dg, dgr, 5e-5)
    _close("ln dbeta", db, dbr, 5e-5)
    rs = _rand((9,), dev, 25).abs() + 0.5  # 333 rows = 9 samples x 37 rows
    dx2, dxa, dg2, db2 = ops.layernorm_bwd_cast(dy, x, meanr, rstdr, g, g_in=gin, rowscale=rs, rows_per_sample=37)
    _, dxar, _, _ = ref.layernorm_bwd_cast(dy, x, meanr, rstdr, g, g_in=gin, rowscale=rs, rows_per_sample=37)
    _close("ln dx (cast variant)", dx2, dxr, 5e-5)
    _close("ln dx_act", dxa, dxar, _tol(dt, bf=8e-3))
    _close("ln dgamma (cast variant)", dg2, dgr, 5e-5)
    # fp32 upstream gradient, only the activation-dtype dx (the patch-embedding norm: no fp32 dx, no cast pass)
    dyf = _rand((rows, C), dev, 26)
    dxa3, dg3, db3 = ops.layernorm_bwd_to_act(dyf, x, meanr, rstdr, g)
    dxr3, dgr3, dbr3 = ref.layernorm_bwd(dyf, x, meanr, rstdr, g)
    assert dxa3.dtype == ops.act_dtype()
    _close("ln dx (fp32 dy -> act)", dxa3, dxr3, _tol(ops.act_dtype(), f32=5e-5, bf=8e-3))
    _close("ln dgamma (fp32 dy -> act)", dg3, dgr3, 5e-5)
    _close("ln dbeta (fp32 dy -> act)", db3, dbr3, 5e-5)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("H,shift", [(24, 3), (6, 3), (14, 0), (3, 0)])
def test_layernorm_window_maps(mods, dt, H, shift):
    ops, ref = mods
    dev = _dev()
    C, nB, ws = 96, 2, 7
    win2tok, tok2win = ops.window_maps(H, H, ws, shift)
    t2w = torch.from_numpy(tok2win).to(dev)
    period = win2tok.size
    x = _rand((nB, H * H, C), dev, 25)
    g, b = _rand((C,), dev, 26) * 0.2 + 1, _rand((C,), dev, 27) * 0.1
    kw = dict(rowmap=t2w, period_out=period, out_rows=nB * period, dtype=dt)
    y, _, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6, **kw)
    yr, _, meanr, rstdr = ref.layernorm_fwd(x, g, b, 1e-6, **kw)
    _close("lnw y", y, yr, _tol(dt, bf=8e-3))
    dy = _rand((nB * period, C), dev, 28, dt)
    dx, dg, db = ops.layernorm_bwd(dy, x, meanr, rstdr, g, rowmap=t2w, period_in=period)
    dxr, dgr, dbr = ref.layernorm_bwd(dy, x, meanr, rstdr, g, rowmap=t2w, period_in=period)
    _close("lnw dx", dx, dxr, 5e-5)
    _close("lnw dgamma", dg, dgr, 5e-5)
    w2t = torch.from_numpy(win2tok).to(dev)
    src = _rand((nB * H * H, C), dev, 29)
    sc = torch.tensor([0.0, 1.25], device=dev)
    kw = dict(rowmap=w2t, tokens=H * H, rowscale=sc, rows_per_sample=H * H, dtype=dt)
    _close("gather_cast", ops.gather_cast(src, nB * period, **kw), ref.gather_cast(src, nB * period, **kw), _tol(dt, bf=8e-3))


@pytest.mark.parametrize("dt", DTYPES)
def test_merge_ln(mods, dt):
    ops, ref = mods
    dev = _dev()
    nB, H, C = 3, 12, 96
    x = _rand((nB, H * H, C), dev, 30)
    g, b = _rand((4 * C,), dev, 31) * 0.2 + 1, _rand((4 * C,), dev, 32) * 0.1
    y, mean, rstd = ops.merge_ln_fwd(x, g, b, 1e-6, H, H, dtype=dt)
    yr, meanr, rstdr = ref.merge_ln_fwd(x, g, b, 1e-6, H, H, dtype=dt)
    _close("merge y", y, yr, _tol(dt, bf=8e-3))
    dy = _rand(tuple(yr.shape), dev, 33, dt)
    dx, dg, db = ops.merge_ln_bwd(dy, x, meanr, rstdr, g, H, H)
    dxr, dgr, dbr = ref.merge_ln_bwd(dy, x, meanr, rstdr, g, H, H)
    _close("merge dx", dx, dxr, 5e-5)
    _close("merge dgamma", dg, dgr, 5e-5)
    _close("merge dbeta", db, dbr, 5e-5)
    # + the cast, row-scaled copy at the un-merged token rows (the shadow of the stage's last block)
    for rs in (_rand((nB * H * H,), dev, 34).abs() + 0.5, None):
        act, actr = torch.empty((nB * H * H, C), dtype=dt, device=dev), torch.empty((nB * H * H, C), dtype=dt, device=dev)
        dx2, dg2, _ = ops.merge_ln_bwd(dy, x, meanr, rstdr, g, H, H, act_out=act, rowscale=rs)
        ref.merge_ln_bwd(dy, x, meanr, rstdr, g, H, H, act_out=actr, rowscale=rs)
        _close("merge dx (shadow variant)", dx2, dxr, 5e-5)
        _close("merge dgamma (shadow variant)", dg2, dgr, 5e-5)
        _close("merge dx_act", act, actr, _tol(dt, bf=8e-3))


@pytest.mark.parametrize("dt", DTYPES)
def test_small_ops(mods, dt):
    ops, ref = mods
    dev = _dev()
    img = _rand((3, 3, 96, 96), dev, 40)
    _close("im2col", ops.patch_im2col(img, 4, 64, dtype=dt), ref.patch_im2col(img, 4, 64, dtype=dt), _tol(dt, bf=8e-3))
    x = _rand((5, 49, 768), dev, 41)
    m, ma = ops.token_mean_fwd(x, dtype=dt)
    mr, mar = ref.token_mean_fwd(x, dtype=dt)
    _close("mean", m, mr, 2e-5)
    _close("mean act", ma, mar, _tol(dt, bf=8e-3))
    gm, gt = _rand((5, 768), dev, 42), _rand((5, 49, 768), dev, 43)
    _close("mean bwd", ops.token_mean_bwd(gm, gt, 49), ref.token_mean_bwd(gm, gt, 49), 2e-5)
    w = _rand((300, 520), dev, 44)
    _close("cast", ops.cast_to_act(w, dtype=dt), ref.cast_to_act(w, dtype=dt), _tol(dt, bf=8e-3))
    xa = _rand((1300, 288), dev, 45, dt)
    _close("colsum", ops.colsum(xa), ref.colsum(xa), 5e-5)
    for rows in (7, 600, 1024):  # few rows: the single-pass form, plain and accumulating into an existing vector
        xb = _rand((rows, 200), dev, 48, torch.float32)
        _close("colsum (few rows)", ops.colsum(xb), ref.colsum(xb), 5e-5)
        base = _rand((200,), dev, 49)
        got = ops.colsum(xb, out=base.clone(), accumulate=True)
        _close("colsum (few rows, accumulate)", got, base + ref.colsum(xb), 5e-5)
    v = _rand((777,), dev, 46)
    _close("sum", ops.sum_f32(v), ref.sum_f32(v), 1e-5)
    xs, xs2 = xa.clone(), xa.clone()
    sc = torch.tensor(0.37, device=dev)
    _close("scale", ops.scale_inplace(xs, sc), ref.scale_inplace(xs2, sc), _tol(dt, bf=8e-3))
    c1 = _rand((1, 4096), dev, 47)
    c2 = c1.clone()
    cs = _rand((4096,), dev, 48)
    _close("center", ops.center_ema(c1, cs, 0.9, 13.0), ref.center_ema(c2, cs, 0.9, 13.0), 1e-6)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("ws,nH,H,shift,hd", [(7, 3, 14, 3, 32), (7, 3, 12, 3, 32), (7, 6, 12, 0, 32), (7, 12, 6, 3, 32), (7, 24, 3, 0, 32),
                                              (7, 6, 7, 0, 32), (14, 3, 28, 7, 32), (14, 3, 24, 7, 32), (14, 6, 12, 7, 32), (14, 4, 14, 0, 32),
                                              (14, 2, 6, 0, 32), (14, 2, 7, 0, 32), (14, 3, 3, 0, 32),
                                              # head_dim 64 (CvT: dim / heads): 7x7 windows, and the 6x6 / 3x3 windows of 96^2 crops
                                              (7, 1, 14, 0, 64), (7, 3, 12, 3, 64), (6, 6, 6, 0, 64), (3, 12, 3, 0, 64), (7, 3, 28, 0, 64)])
def test_window_attention(mods, dt, ws, nH, H, shift, hd):
    """token-ordered attention over every padded / shifted geometry class of 224 and 96 crops, 7x7 and 14x14 windows"""
    ops, ref = mods
    dev = _dev()
    N = ws * ws
    C = nH * hd
    nB, L = 3, H * H
    win2tok_np, _ = ops.window_maps(H, H, ws, shift)
    w2t = torch.from_numpy(win2tok_np).to(dev)
    nW = w2t.numel() // N
    qkv = _rand((nB * L, 3 * C), dev, 50, dt)
    qb = _rand((3 * C,), dev, 49) * 0.5
    trows = (2 * ws - 1) ** 2
    table = _rand((trows, nH), dev, 51) * 0.5
    index = torch.from_numpy(ops.relative_position_index(ws)).to(dev)
    mask_frag = None
    if shift:
        ids_np = ops.shift_region_ids(H, H, ws, shift)
        assert np.array_equal(ids_np, ref.shift_region_ids(H, H, ws, shift))
        mask_frag = torch.from_numpy(ids_np).to(dev)  # the kernels rebuild the 0/-100 mask from the region labels
        m_np = ops.shift_mask(H, H, ws, shift)
        assert np.array_equal(np.where(ids_np.reshape(nW, N, 1) == ids_np.reshape(nW, 1, N), 0.0, -100.0).astype(np.float32), m_np)
    scale = hd ** -0.5
    o, lse, attn = ops.window_attn_fwd(qkv, qb, w2t, L, table, ws, mask_frag, nW, N, nH, scale, want_attn=True)
    orf, _, attnr = ref.window_attn_fwd(qkv, qb, w2t, L, table, ws, mask_frag, nW, N, nH, scale, want_attn=True)
    _close("attn probs", attn, attnr, _tol(dt, f32=5e-5, bf=2e-2))
    _close("attn out", o, orf, _tol(dt, f32=5e-5, bf=2e-2))
    res = ops.window_attn_fwd(qkv, qb, w2t, L, table, ws, mask_frag, nW, N, nH, scale)  # training variant: no probabilities written
    _close("attn out (train)", res[0], orf, _tol(dt, f32=5e-5, bf=2e-2))
    if ws == 14:
        # the training variant skips query tiles (16 slots) without a live token -- the backward skips the same tiles -- and writes 0 there
        live = torch.zeros((nW, 224), dtype=torch.bool, device=dev)
        live[:, :N] = w2t.view(nW, N) >= 0
        live = live.view(nW, 14, 16).any(-1, keepdim=True).expand(nW, 14, 16).reshape(1, nW, 1, 224).expand(nB, nW, nH, 224).reshape(res[1].shape)
        _close("attn lse (train)", res[1], torch.where(live, lse, torch.zeros_like(lse)), _tol(dt, f32=5e-5, bf=2e-2))
    dout = _rand((nB * L, C), dev, 52, dt)
    dqkv, ws_, pad_ = ops.window_attn_bwd(qkv, qb, w2t, L, dout, orf, lse, table, ws, mask_frag, nW, N, nH, scale)
    dt_ = ops.relpos_bias_bwd(ws_, index, N, trows)
    dqkvr, wsr, padr = ref.window_attn_bwd(qkv, qb, w2t, L, dout, orf, lse, table, ws, mask_frag, nW, N, nH, scale)
    for i, nm in enumerate("qkv"):
        _close("attn d%s" % nm, dqkv.view(-1, 3, C)[:, i], dqkvr.view(-1, 3, C)[:, i], _tol(dt, f32=1e-4, bf=3e-2))
    dtr = ref.relpos_bias_bwd(wsr, index, N, trows)
    _close("attn dtable", dt_, dtr, _tol(dt, f32=1e-4, bf=3e-2))
    if (win2tok_np < 0).any():
        _close("attn dpad", pad_.sum(0, keepdim=True), padr, _tol(dt, f32=1e-4, bf=3e-2))
    else:
        assert float(pad_.abs().max()) == 0.0


def _attn_branch_unfused(ops, x, g1, b1, Wqkv, bqkv, Wproj, bproj, w2t, L, table, ws, regions, nW, N, nH, scale, rowscale):
    """the four-kernel sequence the fused branch replaces (functional._block_forward_multi), bf16 activations"""
    xw, _, mean, rstd = ops.layernorm_fwd(x, g1, b1, 1e-6)
    qkv = ops.linear_fwd(xw, Wqkv.to(torch.bfloat16), bqkv)
    ao, _ = ops.window_attn_fwd(qkv, bqkv, w2t, L, table, ws, regions, nW, N, nH, scale)
    y = ops.linear_fwd(ao, Wproj.to(torch.bfloat16), bproj, residual=x, rowscale=rowscale, rows_per_sample=1, out_f32=True)
    return y, xw, mean, rstd, qkv, ao


@pytest.mark.parametrize("nH,H,shift,nB", [(3, 14, 3, 3), (3, 12, 3, 5), (6, 12, 0, 2), (3, 56, 0, 2), (6, 28, 3, 3), (3, 24, 3, 9), (6, 7, 0, 1),
                                           (3, 56, 3, 24), (6, 28, 3, 40), (6, 12, 3, 300)])
@pytest.mark.parametrize("save", [False, True])
def test_attn_branch_fwd(mods, nH, H, shift, nB, save):
    """LayerNorm -> qkv -> 7x7 window attention -> proj -> DropPath-scaled residual in one kernel vs the unfused kernel sequence (same
    bf16 rounding points) and vs the fp32 restatement, over padded / shifted geometries, few and many windows per workgroup"""
    ops, ref = mods
    old = ops.act_dtype()
    ops.set_act_dtype(torch.bfloat16)
    try:
        dev = _dev()
        ws, hd = 7, 32
        N, C, L = ws * ws, nH * hd, H * H
        w2t = torch.from_numpy(ops.window_maps(H, H, ws, shift)[0]).to(dev)
        nW = w2t.numel() // N
        regions = torch.from_numpy(ops.shift_region_ids(H, H, ws, shift)).to(dev) if shift else None
        x = _rand((nB * L, C), dev, 60) + 0.1 * _rand((1, C), dev, 61)
        g1, b1 = 1.0 + 0.1 * _rand((C,), dev, 62), 0.1 * _rand((C,), dev, 63)
        Wqkv, bqkv = _rand((3 * C, C), dev, 64) * C ** -0.5, _rand((3 * C,), dev, 65) * 0.5
        Wproj, bproj = _rand((C, C), dev, 66) * C ** -0.5, _rand((C,), dev, 67) * 0.5
        table = _rand(((2 * ws - 1) ** 2, nH), dev, 68) * 0.5
        keep = (torch.rand(nB, generator=torch.Generator().manual_seed(69)) > 0.3).float() / 0.7
        rowscale = keep.repeat_interleave(L).to(dev)
        scale = hd ** -0.5
        yr, xwr, meanr, rstdr, qkvr, aor = _attn_branch_unfused(ops, x, g1, b1, Wqkv, bqkv, Wproj, bproj, w2t, L, table, ws, regions, nW, N, nH, scale, rowscale)
        Wq_p, Wp_p = ops.cast_weight(Wqkv, perm32=True), ops.cast_weight(Wproj, perm32=True)
        res = ops.attn_branch_fwd(x, g1, b1, 1e-6, Wq_p, bqkv, Wp_p, bproj, w2t, L, table, ws, regions, nW, N, nH, scale, rowscale=rowscale, save=save)
        y = res[0] if save else res
        assert bool(torch.isfinite(y).all())
        _close("branch delta", y - x, yr - x, 2.5e-2)  # two bf16 pipelines with different summation orders
        # fp32 restatement of the same branch
        xw32 = torch.nn.functional.layer_norm(x, (C,), g1, b1, 1e-6)
        qkv32 = xw32 @ Wqkv.t() + bqkv
        ao32 = ref.window_attn_fwd(qkv32, bqkv, w2t, L, table, ws, regions, nW, N, nH, scale)[0]
        y32 = x + rowscale[:, None] * (ao32 @ Wproj.t() + bproj)
        _close("branch delta vs fp32", y - x, y32 - x, 2.5e-2)
        if save:
            xw, mean, rstd, qkv, ao = res[1]
            _close("xw", xw, xwr, 1e-2)
            _close("mean", mean, meanr, 1e-5)
            _close("rstd", rstd, rstdr, 1e-5)
            _close("qkv", qkv, qkvr, 1.5e-2)
            _close("ao", ao, aor, 2e-2)
        # no DropPath vector, prefilled bias fragment, output into a given tensor
        frag = ops.new_bias_frag(nH, N, dev)
        ops.attn_branch_fwd(x, g1, b1, 1e-6, Wq_p, bqkv, Wp_p, bproj, w2t, L, table, ws, regions, nW, N, nH, scale, bias_frag=frag)
        out = torch.empty_like(x)
        ops.attn_branch_fwd(x, g1, b1, 1e-6, Wq_p, bqkv, Wp_p, bproj, w2t, L, None, ws, regions, nW, N, nH, scale, bias_frag=frag, out=out)
        y1 = x + (ao32 @ Wproj.t() + bproj)
        _close("branch delta, no rowscale", out - x, y1 - x, 2.5e-2)
    finally:
        ops.set_act_dtype(old)


@pytest.mark.parametrize("nH,H,shift,nB", [(3, 56, 3, 48), (6, 28, 0, 64)])
def test_attn_branch_fwd_is_bit_reproducible(mods, nH, H, shift, nB):
    """the fused branch at full occupancy, 25 launches on the same inputs: every output (with and without side outputs) identical to
    the bit.  Its cross-lane reductions are hand-written v_permlane16/32_swap sequences; the form that first shipped (the builtin with
    too few wait states behind it) gave run-to-run different results in another kernel"""
    ops, _ = mods
    dev = _dev()
    ws, hd = 7, 32
    N, C, L = ws * ws, nH * hd, H * H
    w2t = torch.from_numpy(ops.window_maps(H, H, ws, shift)[0]).to(dev)
    nW = w2t.numel() // N
    regions = torch.from_numpy(ops.shift_region_ids(H, H, ws, shift)).to(dev) if shift else None
    x = _rand((nB * L, C), dev, 70)
    g1, b1 = 1.0 + 0.1 * _rand((C,), dev, 71), 0.1 * _rand((C,), dev, 72)
    Wq_p, bqkv = ops.cast_weight(_rand((3 * C, C), dev, 73) * C ** -0.5, perm32=True), _rand((3 * C,), dev, 74) * 0.5
    Wp_p, bproj = ops.cast_weight(_rand((C, C), dev, 75) * C ** -0.5, perm32=True), _rand((C,), dev, 76) * 0.5
    table = _rand(((2 * ws - 1) ** 2, nH), dev, 77) * 0.5
    ref = None
    for rep in range(25):
        y, side = ops.attn_branch_fwd(x, g1, b1, 1e-6, Wq_p, bqkv, Wp_p, bproj, w2t, L, table, ws, regions, nW, N, nH, hd ** -0.5, save=True)
        y2 = ops.attn_branch_fwd(x, g1, b1, 1e-6, Wq_p, bqkv, Wp_p, bproj, w2t, L, table, ws, regions, nW, N, nH, hd ** -0.5)
        cur = [y, y2] + list(side)
        assert torch.equal(y, y2), "side outputs change the branch output (rep %d)" % rep
        if ref is None:
            ref = [t.clone() for t in cur]
        else:
            for i, (a, b) in enumerate(zip(cur, ref)):
                assert torch.equal(a, b), "tensor %d differs between launch 0 and launch %d" % (i, rep)


@pytest.mark.parametrize("ws,H,shift,nH,nB", [(7, 56, 3, 3, 48), (7, 14, 3, 12, 160), (14, 28, 7, 6, 48), (14, 12, 0, 12, 96)])
def test_window_attention_is_bit_reproducible(mods, ws, H, shift, nH, nB):
    """the 7x7 and 14x14 window-attention forward and backward (bf16) at full occupancy, 12 launches each on the same inputs: every
    output identical to the bit.  Their 16-byte row stores pack two accumulator tiles through the v_permlane16_swap BUILTIN
    (common.h: esvit_pack_tile_pair_bf16; the integer uses the hazard recogniser pads itself) -- the float butterflies that gave
    run-to-run differences in round 5 were a different use of the instruction (common.h: swap16 / swap32) and live in the fused
    branch only.  This is the check that the store path of these kernels is deterministic under load."""
    ops, _ = mods
    dev = _dev()
    old = ops.act_dtype()
    ops.set_act_dtype(torch.bfloat16)
    try:
        _window_attention_repro(ops, dev, ws, H, shift, nH, nB)
    finally:
        ops.set_act_dtype(old)


def _window_attention_repro(ops, dev, ws, H, shift, nH, nB):
    hd = 32
    N, C, L = ws * ws, nH * hd, H * H
    w2t = torch.from_numpy(ops.window_maps(H, H, ws, shift)[0]).to(dev)
    nW = w2t.numel() // N
    regions = torch.from_numpy(ops.shift_region_ids(H, H, ws, shift)).to(dev) if shift else None
    qkv = _rand((nB * L, 3 * C), dev, 80, torch.bfloat16)
    qb = _rand((3 * C,), dev, 81) * 0.5
    table = _rand(((2 * ws - 1) ** 2, nH), dev, 82) * 0.5
    dout = _rand((nB * L, C), dev, 83, torch.bfloat16)
    first = None
    for rep in range(12):
        o, lse = ops.window_attn_fwd(qkv, qb, w2t, L, table, ws, regions, nW, N, nH, hd ** -0.5)
        dqkv, dbias_ws, dpad_ws = ops.window_attn_bwd(qkv, qb, w2t, L, dout, o, lse, table, ws, regions, nW, N, nH, hd ** -0.5)
        cur = [o, dqkv, dbias_ws, dpad_ws] + ([lse] if lse is not None else [])
        assert all(bool(torch.isfinite(t.float()).all()) for t in cur)
        if first is None:
            first = [t.clone() for t in cur]
        else:
            for i, (a, b) in enumerate(zip(cur, first)):
                assert torch.equal(a, b), "tensor %d differs between launch 0 and launch %d" % (i, rep)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("geom", ["vit37_h3", "vit37_h6", "cvt56_h1", "cvt28_h3", "swin56_h3_hd32", "w14map6_h12_hd32", "w14map24_h3_hd32"])
def test_window_attention_full_occupancy(mods, dt, geom):
    """head_dim 64 (and 32 as the control) with MORE window-heads than the chip keeps resident: every CU fully occupied and
    several windows per workgroup.  Regression for the bench lines of the ViT / CvT / ViL configurations, whose bf16 forward
    stored 0x7ffffff0 (two bf16 NaNs) into four rows of some windows at this occupancy (common.h: buffer_store_b128); the
    unit geometries of test_window_attention (<= 48 window-heads) never showed it."""
    ops, ref = mods
    import esvit_amd.functional as Fn
    dev = _dev()
    vit = geom.startswith("vit")
    hd = 32 if geom.endswith("hd32") else 64
    nH = int(geom.split("_h")[1].split("_")[0])
    C = nH * hd
    if vit:
        N = L = 37
        nB, nW = 1280, 1
        w2t, ws, table = Fn._vit_window(N, nH, dev)
    else:
        # (w14map*: 14 x 14 windows that are mostly padding -- the 6 x 6 / 24 x 24 maps of the 96^2 crops; the three kernels skip the query
        # tiles without a live slot there, per window)
        H, ws, nB = {"cvt56": (56, 7, 40), "swin56": (56, 7, 40), "cvt28": (28, 7, 64), "w14map6": (6, 14, 96), "w14map24": (24, 14, 64)}[geom.split("_")[0]]
        N, L = ws * ws, H * H
        w2t = torch.from_numpy(ops.window_maps(H, H, ws, 0)[0]).to(dev)
        nW = w2t.numel() // N
        table = _rand(((2 * ws - 1) ** 2, nH), dev, 51) * 0.5
    qkv = _rand((nB * L, 3 * C), dev, 50, dt)
    qb = _rand((3 * C,), dev, 49) * 0.5
    dout = _rand((nB * L, C), dev, 52, dt)
    scale = hd ** -0.5
    junk = [torch.full((1 << 26,), float("nan"), device=dev) for _ in range(8)]  # (what torch.empty hands out next is poisoned)
    del junk
    for rep in range(2):
        o, lse = ops.window_attn_fwd(qkv, qb, w2t, L, table, ws, None, nW, N, nH, scale)
        dqkv = ops.window_attn_bwd(qkv, qb, w2t, L, dout, o, lse, table, ws, None, nW, N, nH, scale)[0]
        assert bool(torch.isfinite(o.float()).all()) and bool(torch.isfinite(dqkv.float()).all())
        step = 160 if vit else (32 if ws == 14 else 4)
        for b0 in range(0, nB, step):
            sl = slice(b0 * L, min(nB, b0 + step) * L)
            orf = ref.window_attn_fwd(qkv[sl], qb, w2t, L, table, ws, None, nW, N, nH, scale)
            _close("attn out, images %d.." % b0, o[sl], orf[0], _tol(dt, f32=5e-5, bf=2e-2))
            dr = ref.window_attn_bwd(qkv[sl], qb, w2t, L, dout[sl], orf[0], orf[1], table, ws, None, nW, N, nH, scale)[0]
            _close("attn dqkv, images %d.." % b0, dqkv[sl], dr, _tol(dt, f32=1e-4, bf=3e-2))


@pytest.mark.parametrize("dt", DTYPES)
def test_cvt_conv_pieces(mods, dt):
    """ConvEmbed im2col / col2im, depthwise 3x3 (+ flipped, + weight gradient), BatchNorm reductions and affine apply"""
    ops, ref = mods
    dev = _dev()
    # stage-0 embed: 7x7 stride 4 pad 2 on NCHW images (K = 147 -> padded to 152)
    img = _rand((3, 3, 40, 40), dev, 70)
    _close("im2col nchw", ops.conv_im2col(img, True, 3, 40, 40, 3, 7, 4, 2, dtype=dt), ref.conv_im2col(img, True, 3, 40, 40, 3, 7, 4, 2, dtype=dt),
           _tol(dt))
    # Vision Longformer's patch embeddings: 4x4 stride 4 on images, 2x2 stride 2 on tokens, no padding
    _close("im2col nchw k4", ops.conv_im2col(img, True, 3, 40, 40, 3, 4, 4, 0, dtype=dt), ref.conv_im2col(img, True, 3, 40, 40, 3, 4, 4, 0, dtype=dt), _tol(dt))
    tok = _rand((2 * 12 * 12, 48), dev, 76, dt)
    _close("im2col nhwc k2", ops.conv_im2col(tok, False, 2, 12, 12, 48, 2, 2, 0), ref.conv_im2col(tok, False, 2, 12, 12, 48, 2, 2, 0), _tol(dt))
    dcols = _rand((2 * 6 * 6, 4 * 48), dev, 77, dt)
    _close("col2im k2", ops.conv_col2im(dcols, 2, 12, 12, 48, 2, 2, 0), ref.conv_col2im(dcols, 2, 12, 12, 48, 2, 2, 0), _tol(dt, f32=1e-5, bf=1e-5))
    # later embeds: 3x3 stride 2 pad 1 on NHWC tokens, odd and even grids
    for H, Cin in ((14, 64), (5, 24)):
        tok = _rand((2 * H * H, Cin), dev, 71, dt)
        _close("im2col nhwc", ops.conv_im2col(tok, False, 2, H, H, Cin, 3, 2, 1), ref.conv_im2col(tok, False, 2, H, H, Cin, 3, 2, 1), _tol(dt))
        Ho = ops.conv_out_size(H, 3, 2, 1)
        dcols = _rand((2 * Ho * Ho, -(-(9 * Cin) // 8) * 8), dev, 72, dt)
        _close("col2im", ops.conv_col2im(dcols, 2, H, H, Cin, 3, 2, 1), ref.conv_col2im(dcols, 2, H, H, Cin, 3, 2, 1), _tol(dt, f32=1e-5, bf=1e-5))
    for H, Cc in ((14, 64), (6, 192), (3, 32)):
        x = _rand((2 * H * H, Cc), dev, 73, dt)
        w = _rand((Cc, 9), dev, 74) * 0.3
        _close("dwconv", ops.dwconv3x3(x, w, 2, H, H), ref.dwconv3x3(x, w, 2, H, H), _tol(dt, bf=2e-2))
        _close("dwconv flip", ops.dwconv3x3(x, w, 2, H, H, flip=True), ref.dwconv3x3(x, w, 2, H, H, flip=True), _tol(dt, bf=2e-2))
        dy = _rand((2 * H * H, Cc), dev, 75, dt)
        _close("dwconv wgrad", ops.dwconv3x3_wgrad(x, dy, 2, H, H), ref.dwconv3x3_wgrad(x, dy, 2, H, H), _tol(dt, f32=1e-4, bf=1e-4))
        _close("col sums (bn stats)", ops.col_sums2(x, x), ref.col_sums2(x, x), _tol(dt, f32=1e-4, bf=1e-4))
        _close("col sums (bn bwd)", ops.col_sums2(dy, x), ref.col_sums2(dy, x), _tol(dt, f32=1e-4, bf=1e-4))
        a1, a2, a3 = _rand((Cc,), dev, 76), _rand((Cc,), dev, 77), _rand((Cc,), dev, 78)
        _close("affine", ops.col_affine2(x, a1, a3), ref.col_affine2(x, a1, a3), _tol(dt, bf=2e-2))
        _close("affine2", ops.col_affine2(x, a1, a3, dy, a2), ref.col_affine2(x, a1, a3, dy, a2), _tol(dt, bf=2e-2))
        _close("affine relu", ops.col_affine2(x, a1, a3, act=3), ref.col_affine2(x, a1, a3, act=3), _tol(dt, bf=2e-2))
        gate, gate_ref = ops.col_affine2(x, a1, a3, dy, act=4), ref.col_affine2(x, a1, a3, dy, act=4)
        assert (gate != gate_ref).sum().item() <= 2, "affine relu bwd"  # (a pre-activation within rounding of zero may gate either way)
        _close("pad", ops.pad_crop_tokens(x, 2, H, H, H + 2, H + 1), ref.pad_crop_tokens(x, 2, H, H, H + 2, H + 1), 0.0)
        _close("crop", ops.pad_crop_tokens(x, 2, H, H, H - 1, H - 2), ref.pad_crop_tokens(x, 2, H, H, H - 1, H - 2), 0.0)
        # BatchNorm coefficient vectors
        sums, gam, bet = ops.col_sums2(x, x), 1 + 0.1 * _rand((Cc,), dev, 79), 0.1 * _rand((Cc,), dev, 80)
        rm, rv = _rand((Cc,), dev, 81) * 0.1, 1 + 0.1 * _rand((Cc,), dev, 82).abs()
        rm2, rv2 = rm.clone(), rv.clone()
        coef, coefr = ops.bn_fwd_coeffs(sums, 2 * H * H, gam, bet, 1e-5, 0.1, rm, rv), ref.bn_fwd_coeffs(sums, 2 * H * H, gam, bet, 1e-5, 0.1, rm2, rv2)
        _close("bn coef", coef, coefr, 1e-4)
        _close("bn running mean", rm, rm2, 1e-5)
        _close("bn running var", rv, rv2, 1e-4)
        _close("bn eval coef", ops.bn_eval_coeffs(rm, rv, gam, bet, 1e-5), ref.bn_eval_coeffs(rm, rv, gam, bet, 1e-5), 1e-5)
        red, redr = ops.bn_bwd_local(ops.col_sums2(dy, x), coef), ref.bn_bwd_local(ref.col_sums2(dy, x), coefr)
        _close("bn bwd local", red, redr, _tol(dt, f32=1e-3, bf=1e-3))
        _close("bn bwd coef", ops.bn_bwd_coeffs(red, 2 * H * H, gam, coef), ref.bn_bwd_coeffs(red, 2 * H * H, gam, coef), 1e-4)
        _close("bn bwd coef eval", ops.bn_bwd_coeffs(None, 2 * H * H, gam, coef), ref.bn_bwd_coeffs(None, 2 * H * H, gam, coef), 1e-4)


@pytest.mark.parametrize("dt", DTYPES)
def test_head_pieces(mods, dt):
    ops, ref = mods
    dev = _dev()
    x = _rand((300, 256), dev, 60, dt)
    z, inv = ops.l2norm_fwd(x)
    zr, invr = ref.l2norm_fwd(x)
    _close("l2 z", z, zr, _tol(dt, bf=8e-3))
    _close("l2 inv", inv, invr, 2e-5)
    dz = _rand((300, 256), dev, 61, dt)
    _close("l2 bwd", ops.l2norm_bwd(dz, zr, invr), ref.l2norm_bwd(dz, zr, invr), _tol(dt, bf=1e-2))
    v, g = _rand((4096, 256), dev, 62) * 0.02, torch.ones((4096, 1), device=dev) * 1.5
    w, winv = ops.weightnorm_fwd(v, g, dtype=dt)
    wr, winvr = ref.weightnorm_fwd(v, g, dtype=dt)
    _close("wn w", w, wr, _tol(dt, bf=8e-3))
    _close("wn inv", winv, winvr, 2e-5)
    dw = _rand((4096, 256), dev, 63)
    dv, dg = ops.weightnorm_bwd(dw, v, g, winvr, True)
    dvr, dgr = ref.weightnorm_bwd(dw, v, g, winvr, True)
    _close("wn dv", dv, dvr, 5e-5)
    _close("wn dg", dg, dgr, 5e-5)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("K", [4096, 65536])
def test_dino_loss_kernels(mods, dt, K):
    ops, ref = mods
    dev = _dev()
    Rs, Rt = 37, 11
    s, t = _rand((Rs, K), dev, 70, dt), _rand((Rt, K), dev, 71, dt)
    center = _rand((1, K), dev, 72) * 0.3
    mx, lse = ops.teacher_row_stats(t, center, 1 / 0.04)
    mxr, lser = ref.teacher_row_stats(t, center, 1 / 0.04)
    _close("t max", mx, mxr, 1e-5)
    _close("t lse", lse, lser, 2e-4)
    g = torch.Generator().manual_seed(73)
    tm = torch.randint(-1, Rt, (Rs, 2), generator=g).to(torch.int32)
    tm[0] = torch.tensor([-1, 3])
    tm[1] = torch.tensor([2, -1])
    tm = tm.to(dev)
    w = (torch.rand(Rs, generator=g) * 0.1).to(dev)
    rl, ds = ops.dino_ce(s, t, center, mxr, lser, tm, w, 10.0, 25.0)
    rlr, dsr = ref.dino_ce(s, t, center, mxr, lser, tm, w, 10.0, 25.0)
    _close("row loss", rl, rlr, 2e-5)
    _close("ds", ds, dsr, _tol(dt, f32=2e-4, bf=1e-2))
    # the work order (row_order, XCD-contiguous ids) changes which workgroup takes which row, never the result
    order = torch.randperm(Rs, generator=g).to(torch.int32).to(dev)
    rl2, ds2 = ops.dino_ce(s, t, center, mxr, lser, tm, w, 10.0, 25.0, row_order=order)
    assert torch.equal(rl2, rl) and torch.equal(ds2, ds)


@pytest.mark.parametrize("M,N,K", [(256, 4096, 64), (640, 8192, 256), (128, 65536, 256), (1024, 65536, 256)])
@pytest.mark.parametrize("centred", [False, True])
def test_last_layer_row_statistics(mods, M, N, K, centred):
    """esvit_gemm_desc::rowstat + esvit_rowstat_combine: the GEMM that writes the logits also returns the softmax statistics of the
    STORED (bf16-rounded) logits -- compared with esvit_teacher_row_stats on the very tensor it wrote (same values, fp32 sums in
    another order: 2e-5), then used as the student statistics of the CE kernel in place of its first pass"""
    ops, ref = mods
    dev = _dev()
    dt = torch.bfloat16
    assert ops.row_stats_supported(dt, M, N) and not ops.row_stats_supported(dt, M + 8, N) and not ops.row_stats_supported(torch.float32, M, N)
    z = torch.nn.functional.normalize(_rand((M, K), dev, 75), dim=1).to(dt)
    w = torch.nn.functional.normalize(_rand((N, K), dev, 76), dim=1).to(dt)
    cen = (_rand((N,), dev, 77) * 0.2) if centred else None
    inv_t = 25.0 if centred else 10.0
    # (the largest case is whole 256 x 256 tiles: it runs on the eight-phase loop, whose statistics come in 32-column blocks)
    ops.FORCE_GEMM_KERNEL = 5 if (M % 256 == 0 and N % 256 == 0 and M >= 1024) else 0
    try:
        assert ops.gemm_select(dt, M=M, N=N, K=K, rowstat=z)[0] == (5 if ops.FORCE_GEMM_KERNEL else 2)
        y, mx, lse = ops.linear_fwd(z, w, row_stats=(inv_t, cen, True))
    finally:
        ops.FORCE_GEMM_KERNEL = 0
    assert torch.equal(y, ops.linear_fwd(z, w))  # the logits themselves do not change
    # esvit_gemm_desc::colstat: the batch sum of the STORED logits per column (the centre update's input) from the same epilogue --
    # the 128 x 128 tile only; the eight-phase loop does not offer it and the attribute is then absent
    cs = getattr(mx, "esvit_col_sums", None)
    if M % 256 == 0 and N % 256 == 0 and M >= 1024:
        assert cs is None
    else:
        assert cs is not None and cs.shape == (N,)
        _close("column sums of the stored logits", cs, y.float().sum(0), 2e-5)
        _close("column sums vs esvit_colsum", cs, ops.colsum(y), 2e-5)
    zero = torch.zeros(N, device=dev)
    mx0, lse0 = ops.teacher_row_stats(y, zero if cen is None else cen, inv_t)
    _close("row max", mx, mx0, 1e-6)
    _close("row lse", mx + lse, mx0 + lse0, 2e-5)
    yr, mxr, lser = ref.linear_fwd(z, w, row_stats=(inv_t, cen))
    _close("logits", y, yr, 1e-2)
    _close("lse vs restatement", mx + lse, mxr + lser, 2e-3)  # (a logit rounded the other way moves the row maximum)
    if not centred:
        Rt = 16
        t = _rand((Rt, N), dev, 78, dt)
        c1 = _rand((1, N), dev, 79) * 0.3
        tmx, tlse = ref.teacher_row_stats(t, c1, 25.0)
        g = torch.Generator().manual_seed(80)
        tm = torch.randint(-1, Rt, (M, 2), generator=g).to(torch.int32).to(dev)
        wr = (torch.rand(M, generator=g) * 0.1).to(dev)
        rl0, ds0 = ops.dino_ce(y, t, c1, tmx, tlse, tm, wr, inv_t, 25.0)
        rl1, ds1 = ops.dino_ce(y, t, c1, tmx, tlse, tm, wr, inv_t, 25.0, s_stats=(mx, lse))
        _close("row loss with handed-over statistics", rl1, rl0, 2e-5)
        _close("ds with handed-over statistics", ds1, ds0, 1e-2)


def test_row_statistics_are_refused_outside_their_epilogue(mods):
    ops, _ = mods
    dev = _dev()
    z, w = _rand((128, 64), dev, 81, torch.bfloat16), _rand((256, 64), dev, 82, torch.bfloat16)
    y = torch.empty((128, 256), dtype=torch.bfloat16, device=dev)
    st = torch.empty((128, 4, 2), device=dev)
    with pytest.raises(RuntimeError, match="row statistics"):
        ops._gemm(torch.bfloat16, A=z, B=w, C=y, M=128, N=256, K=64, lda=64, ldb=64, ldc=256, rowstat=st, rowstat_scale=1.0, bias=torch.zeros(256, device=dev))
    with pytest.raises(RuntimeError, match="row statistics"):
        ops._gemm(torch.bfloat16, A=z, B=w, C=y, M=120, N=256, K=64, lda=64, ldb=64, ldc=256, rowstat=st, rowstat_scale=1.0)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("grid,w,nglo,hd", [((28, 28), 7, 1, 48), ((12, 12), 7, 1, 32), ((14, 14), 7, 2, 64), ((56, 56), 7, 1, 48)])
def test_sliding_chunk_attention(mods, dt, grid, w, nglo, hd):
    """Vision Longformer's chunk-neighbourhood attention (esvit_softmax_rows_chunked_fwd on the batched-GEMM route) forward and
    backward vs the restatement, whose mask is checked against the reference's sliding-chunk implementation by the ViL fixtures"""
    ops, ref = mods
    dev = _dev()
    nx, ny = grid
    N, B, nH = nglo + nx * ny, 2, 2
    ix, iy = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    chunk = torch.from_numpy(np.concatenate([np.full(nglo, -1), ((ix // w) << 16 | (iy // w)).reshape(-1)]).astype(np.int32)).to(dev)
    qkv = _rand((B * N, 3 * nH * hd), dev, 90, dt)
    scale = hd ** -0.5
    out, saved = ops.vit_attn_fwd(qkv, B, N, nH, scale, chunk=chunk)
    outr, savedr = ref.vit_attn_fwd(qkv, B, N, nH, scale, chunk=chunk)
    # with the token order declared (chunk row by chunk row) the kernels skip what a query cannot see: same result up to the
    # order of the partial sums (lanes start at another column)
    lay = (chunk, nglo, w * ny)
    out2, saved2 = ops.vit_attn_fwd(qkv, B, N, nH, scale, chunk=lay)
    _close("span vs full scan: out", out2, out, _tol(dt, f32=2e-6, bf=8e-3))
    _close("span vs full scan: P", saved2[1], saved[1], _tol(dt, f32=2e-6, bf=8e-3))
    _close("sliding-chunk out", out, outr, _tol(dt, f32=2e-5, bf=2e-2))
    p = saved[1].float().view(B * nH, saved[1].shape[-2], saved[1].shape[-1])[:, :N, :N]
    allowed = ref.chunk_mask(chunk).to(dev)
    assert (p[:, ~allowed] == 0).all() and abs(p.sum(-1).mean().item() - 1.0) < 2e-2  # nothing outside the neighbourhood, rows sum to one
    dout = _rand((B * N, nH * hd), dev, 91, dt)
    dq = ops.vit_attn_bwd(dout, saved, B, N, nH, scale)
    _close("sliding-chunk dqkv", dq, ref.vit_attn_bwd(dout, savedr, B, N, nH, scale), _tol(dt, f32=5e-5, bf=3e-2))
    _close("span vs full scan: dqkv", ops.vit_attn_bwd(dout, saved, B, N, nH, scale, chunk=lay), dq, _tol(dt, f32=2e-6, bf=8e-3))


def test_index_maps_match_restatement(mods):
    ops, ref = mods
    for ws in (7, 14):
        assert np.array_equal(ops.relative_position_index(ws), ref.relative_position_index(ws))
        for H in (56, 28, 14, 7, 24, 12, 6, 3):
            for shift in (0, ws // 2):
                a, b = ops.window_maps(H, H, ws, shift), ref.window_maps(H, H, ws, shift)
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (ws, H, shift)
            assert np.array_equal(ops.shift_mask(H, H, ws, ws // 2), ref.shift_mask(H, H, ws, ws // 2)), (ws, H)


def test_fused_update_matches_torch(mods):
    """clip (utils.py:106-115) + torch.optim.AdamW + EMA (main_esvit.py:587-590) for 3 steps."""
    import copy
    from esvit_amd.update import FusedClipAdamWEMA, get_params_groups
    dev = _dev()
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(300, 77), torch.nn.LayerNorm(77), torch.nn.Linear(77, 5000)).to(dev)
    net[2].last_layer = None
    net[0].weight.requires_grad_(True)
    frozen = torch.nn.Parameter(torch.ones(10, device=dev), requires_grad=False)
    net.register_parameter("frozen", frozen)
    s_ref, t_ref = copy.deepcopy(net), copy.deepcopy(net)
    s_fus, t_fus = copy.deepcopy(net), copy.deepcopy(net)
    for t in (t_ref, t_fus):
        for p in t.parameters():
            p.data.mul_(0.5)
    opt_ref = torch.optim.AdamW(get_params_groups(s_ref))
    opt_fus = FusedClipAdamWEMA(s_fus, t_fus)
    for it in range(3):
        lr, wd, m = 1e-3 * (it + 1), 0.04 + 0.01 * it, 0.99
        grads = [torch.randn_like(p) * (10.0 if i == 0 else 0.01) for i, p in enumerate(s_ref.parameters())]
        for (p1, p2, g) in zip(s_ref.parameters(), s_fus.parameters(), grads):
            if p1.requires_grad:
                p1.grad, p2.grad = g.clone(), g.clone()
        # reference
        for i, pg in enumerate(opt_ref.param_groups):
            pg["lr"] = lr
            if i == 0:
                pg["weight_decay"] = wd
        for p in s_ref.parameters():
            if p.grad is not None:
                n = p.grad.norm(2)
                coef = 3.0 / (n + 1e-6)
                if coef < 1:
                    p.grad.mul_(coef)
        opt_ref.step()
        with torch.no_grad():
            for q, k in zip(s_ref.parameters(), t_ref.parameters()):
                k.mul_(m).add_((1 - m) * q.detach())
        opt_fus.step(lr, wd, m, clip_grad=3.0)
    for (a, b) in zip(s_ref.parameters(), s_fus.parameters()):
        _close("student param", b, a, 2e-5)
    for (a, b) in zip(t_ref.parameters(), t_fus.parameters()):
        _close("teacher param", b, a, 2e-5)
    sd_ref, sd_fus = opt_ref.state_dict(), opt_fus.state_dict()
    assert [g["params"] for g in sd_ref["param_groups"]] == [g["params"] for g in sd_fus["param_groups"]]
    for k in sd_ref["state"]:
        _close("exp_avg", sd_fus["state"][k]["exp_avg"], sd_ref["state"][k]["exp_avg"], 2e-5)
        _close("exp_avg_sq", sd_fus["state"][k]["exp_avg_sq"], sd_ref["state"][k]["exp_avg_sq"], 2e-5)
        assert float(sd_fus["state"][k]["step"]) == float(sd_ref["state"][k]["step"])


@pytest.mark.parametrize("rule", ["adamw", "sgd", "lars"])
def test_fused_update_skips_non_finite_step(mods, rule):
    """a NaN / inf gradient anywhere makes the whole fused update a no-op (the reference exits before its update on a non-finite
    loss, main_esvit.py:546-551): student, optimizer state and teacher keep their values; the next finite step updates again"""
    import copy
    from esvit_amd.update import FusedClipAdamWEMA
    dev = _dev()
    torch.manual_seed(1)
    net = torch.nn.Sequential(torch.nn.Linear(64, 33), torch.nn.LayerNorm(33), torch.nn.Linear(33, 700)).to(dev)
    teacher = copy.deepcopy(net)
    for p in teacher.parameters():
        p.data.mul_(0.5)
        p.requires_grad_(False)
    kw = {} if rule == "adamw" else dict(rule=rule, momentum=0.9)
    opt = FusedClipAdamWEMA(net, teacher, **kw)

    def step(poison):
        for i, p in enumerate(net.parameters()):
            p.grad = torch.randn_like(p)
        if poison is not None:
            list(net.parameters())[2].grad.view(-1)[5] = poison
        opt.step(1e-2, 0.05, 0.9, clip_grad=3.0)
        torch.cuda.synchronize()

    step(None)
    assert opt.take_skipped() == 0 and set(opt.steps) == {1}
    snap = [p.detach().clone() for p in list(net.parameters()) + list(teacher.parameters()) + list(opt.exp_avg)]
    for bad in (float("nan"), float("inf")):
        step(bad)
        now = list(net.parameters()) + list(teacher.parameters()) + list(opt.exp_avg)
        assert all(torch.equal(a, b.detach()) for a, b in zip(snap, now)), "a non-finite step changed the state"
    # the refused updates are counted on the device and come back out of the host's step counts
    assert set(opt.steps) == {3} and opt.take_skipped() == 2 and set(opt.steps) == {1} and opt.take_skipped() == 0
    step(None)
    assert not torch.equal(snap[0], list(net.parameters())[0].detach()) and set(opt.steps) == {2}
