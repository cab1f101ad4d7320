"""Tensor-level wrappers over the C-ABI of libesvit_hip.so.

Each function takes/returns torch CUDA tensors, allocates outputs with torch (the library never
allocates) and enqueues on torch's current stream.  There is no CPU or eager fallback: a
non-CUDA tensor is an error.  Shapes and semantics are documented in include/esvit_hip.h.
"""
import ctypes as C
import os

import numpy as np
import torch

from ._lib import BF16, EPI_GELU, EPI_GELU_BWD, EPI_NONE, EPI_QGELU, EPI_QGELU_BWD, F32, GEMM_AUTO, GEMM_DMA8, GEMM_P8, GemmDesc, check, lib

_ACT_DTYPE = torch.bfloat16


def set_act_dtype(dt):
    """Storage type of activations: torch.bfloat16 (benchmark mode) or torch.float32 (exact-parity mode)."""
    global _ACT_DTYPE
    assert dt in (torch.bfloat16, torch.float32)
    _ACT_DTYPE = dt


def act_dtype():
    return _ACT_DTYPE


def _code(dt):
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float32:
        return F32
    raise TypeError("unsupported activation dtype %s" % dt)


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("esvit_amd.ops: tensor is not on the GPU (no CPU fallback exists)")
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32c(t):
    assert t.dtype == torch.float32 and t.is_contiguous(), (t.dtype, t.is_contiguous())
    return t


def _actc(t):
    assert t.dtype in (torch.float32, torch.bfloat16) and t.is_contiguous()
    return t


# ------------------------------------------------------------------------------------------------
# shared scratch (split-K partials, column-sum partials, LN partials); stream-ordered reuse
# ------------------------------------------------------------------------------------------------
_WS = {}


def workspace(nfloats, device, slot=0):
    # stream-ordered reuse: one scratch per (device, slot, stream) -- work enqueued on different streams never shares it
    key = (device, slot, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nfloats:
        buf = torch.empty(max(int(nfloats), 1 << 20), dtype=torch.float32, device=device)
        _WS[key] = buf
    return buf


# ------------------------------------------------------------------------------------------------
# host-side integer index maps
# ------------------------------------------------------------------------------------------------
def relative_position_index(ws):
    out = np.empty((ws * ws, ws * ws), dtype=np.int64)
    check(lib.esvit_relative_position_index(ws, out.ctypes.data_as(C.c_void_p)), "relative_position_index")
    return out


# esvit_query questions (include/esvit_hip.h)
(Q_ATTN_FRAG_ELEMS, Q_ATTN_LSE_ELEMS, Q_ATTN_BWD_PARTS, Q_ATTN_BWD_PAD_ROWS, Q_LN_BWD_BLOCKS, Q_COLSUM_BLOCKS, Q_COL_REDUCE_BLOCKS,
 Q_UPDATE_CHUNK_ELEMS, Q_MLP_FUSED, Q_AUG_MAX_BOX) = range(1, 11)


def query(what, a=0, b=0, c=0):
    """scratch sizes / capabilities of the library (esvit_query)"""
    return int(lib.esvit_query(int(what), int(a), int(b), int(c)))


def window_maps(H, W, ws, shift):
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    win2tok = np.empty((Hp // ws) * (Wp // ws) * ws * ws, dtype=np.int32)
    tok2win = np.empty(H * W, dtype=np.int32)
    check(lib.esvit_window_maps(H, W, ws, shift, win2tok.ctypes.data_as(C.c_void_p), tok2win.ctypes.data_as(C.c_void_p), None),
          "window_maps")
    return win2tok, tok2win


def shift_mask(H, W, ws, shift):
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    nW, N = (Hp // ws) * (Wp // ws), ws * ws
    mask = np.empty((nW, N, N), dtype=np.float32)
    n = C.c_int(0)
    check(lib.esvit_shift_mask(H, W, ws, shift, mask.ctypes.data_as(C.c_void_p), C.byref(n)), "shift_mask")
    assert n.value == nW
    return mask


def shift_region_ids(H, W, ws, shift):
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    ids = np.empty((Hp // ws) * (Wp // ws) * ws * ws, dtype=np.int32)
    check(lib.esvit_window_maps(H, W, ws, shift, None, None, ids.ctypes.data_as(C.c_void_p)), "window_maps(region ids)")
    return ids


# ------------------------------------------------------------------------------------------------
# GEMM family
# ------------------------------------------------------------------------------------------------
# optional launch timing for bench.py's roofline leg: list of (flops, start_event, end_event) or None
GEMM_PROFILE = None
_EVENT_POOL = []


def _event():
    return _EVENT_POOL.pop() if _EVENT_POOL else torch.cuda.Event(enable_timing=True)


def _gemm(dt, **kw):
    if GEMM_PROFILE is not None:
        e0, e1 = _event(), _event()
        e0.record()
        _gemm_launch(dt, **kw)
        e1.record()
        GEMM_PROFILE.append((2.0 * kw["M"] * kw["N"] * kw["K"] * kw.get("batch", 1), e0, e1,
                             (kw["M"], kw["N"], kw["K"], kw.get("a_kstrided", 0), kw.get("b_kstrided", 0), kw.get("splitk", 0)),
                             gemm_algorithmic_bytes(dt, kw)))
        return
    _gemm_launch(dt, **kw)


def gemm_algorithmic_bytes(dt, kw):
    """HBM bytes one esvit_gemm call has to move when every operand is read once and every output written once:
    A + B + C (+ the GELU pre-activation / derivative side tensor, + the fp32 residual).  Split-K partials are NOT
    algorithmic (they are this implementation's overhead and show up in the PMC traffic instead)."""
    es = 2 if dt == torch.bfloat16 else 4
    M, N, K, batch = kw["M"], kw["N"], kw["K"], kw.get("batch", 1)
    b = batch * (M * K + N * K) * es + batch * M * N * (4 if kw.get("out_f32", 0) else es)
    if kw.get("aux") is not None:
        b += M * N * es
    if kw.get("residual") is not None:
        b += M * N * 4
    return float(b)


# main loop forced on every esvit_gemm call issued through this module (GEMM_AUTO = the library's own choice); the
# descriptor carries it, the library keeps no state.  Tests sweep it to run every main loop on every shape.
FORCE_GEMM_KERNEL = GEMM_AUTO


def _gemm_desc(kw):
    d = GemmDesc()
    for k in ("A", "B", "C", "bias", "residual", "rowmap", "rowscale", "aux", "partial", "colsum", "colsum_partial", "rowstat", "rowstat_center", "colstat"):
        setattr(d, k, _p(kw.get(k)))
    d.rowstat_scale = float(kw.get("rowstat_scale", 0.0))
    for k in ("M", "N", "K", "lda", "ldb", "ldc", "a_kstrided", "b_kstrided", "strideA", "strideB", "strideC", "ldr",
              "rowmap_period", "rowmap_tokens", "rows_per_sample", "ldaux", "epilogue", "out_f32", "splitk", "accumulate"):
        setattr(d, k, int(kw.get(k, 0)))
    d.batch = int(kw.get("batch", 1))
    d.alpha = float(kw.get("alpha", 1.0))
    d.kernel = int(kw.get("kernel", FORCE_GEMM_KERNEL))
    if d.kernel == GEMM_DMA8 and d.a_kstrided and "kernel" not in kw:
        d.kernel = GEMM_AUTO  # FORCE_GEMM_KERNEL is a test / bench hook: the 8-wave tile has no weight-gradient instantiation
    if d.kernel == GEMM_P8 and "kernel" not in kw:
        # (same hook) the eight-phase loops run whole 64-deep k-tiles, have no row map / statistics, and the 256 x 128 one its own list
        # of epilogue kinds: ask the library whether the forced loop takes this descriptor
        tm, tn, slots = C.c_int(0), C.c_int(0), C.c_int(0)
        if lib.esvit_gemm_select(BF16, C.byref(d), C.byref(tm), C.byref(tn), C.byref(slots)) <= 0:
            d.kernel = GEMM_AUTO
    return d


def _gemm_launch(dt, **kw):
    d = _gemm_desc(kw)
    if dt != torch.bfloat16:
        d.kernel = GEMM_AUTO  # the exact-fp32 mode has one main loop
    check(lib.esvit_gemm(_code(dt), C.byref(d), _stream()), "esvit_gemm")


def gemm_select(dt, **kw):
    """-> (kernel id, tile_m, tile_n, resident workgroup slots) the library would use for this problem (no pointers needed)"""
    d = _gemm_desc(kw)
    if dt != torch.bfloat16:
        d.kernel = GEMM_AUTO
    tm, tn, slots = C.c_int(0), C.c_int(0), C.c_int(0)
    k = lib.esvit_gemm_select(_code(dt), C.byref(d), C.byref(tm), C.byref(tn), C.byref(slots))
    if k <= 0:
        raise RuntimeError("esvit_gemm_select failed (%d): %s" % (k, lib.esvit_last_error().decode()))
    return k, tm.value, tn.value, slots.value


LOG2E = 1.4426950408889634


def row_stats_supported(dt, M, N):
    """can the GEMM that writes logits [M, N] also emit their softmax row statistics (esvit_gemm_desc::rowstat)?"""
    return dt == torch.bfloat16 and M > 0 and M % 128 == 0 and N % 128 == 0


def linear_fwd(x, w, bias=None, *, gelu=False, want_preact=False, residual=None, rowmap=None, rowmap_tokens=0,
               out_rows=None, rowscale=None, rows_per_sample=0, out_f32=False, quick=False, row_stats=None):
    """y = x @ w^T (+bias) with the fused epilogue of the GEMM kernel.  gelu: exact erf-GELU, or (quick=True) the
    QuickGELU x*sigmoid(1.702x) of the CvT feed-forward.

    x [M, K] act; w [N, K] act (cached cast of the fp32 parameter); bias fp32 [N].
    rowmap (int32 [period]) scatters window rows to token rows (out_rows rows, tokens per image =
    rowmap_tokens); residual fp32 [out_rows, N] is added at the destination row.
    row_stats = (inv_temp, center fp32 [N] or None[, want_col_sums]): also return the softmax statistics of z = (y - center) * inv_temp,
    the outputs of teacher_row_stats -> (y, row_max, row_lse); the shape must pass row_stats_supported.  With want_col_sums the batch
    sum of the stored logits per column (what the centre update reads, main_esvit.py:752-770) rides along as the attribute
    `esvit_col_sums` of row_max (fp32 [N]; absent when the main loop in use has no such epilogue)."""
    x, w = _actc(x), _actc(w)
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and x.dtype == w.dtype
    if row_stats is not None:
        inv_temp, cen = row_stats[:2]
        want_cs = len(row_stats) > 2 and bool(row_stats[2])
        assert row_stats_supported(x.dtype, M, N) and bias is None and not gelu and residual is None and rowmap is None and not out_f32
        y = torch.empty((M, N), dtype=x.dtype, device=x.device)
        cen = None if cen is None else _f32c(cen)
        # the statistics come in blocks of 64 columns from the 128-row loop, of 32 columns from the 256 x 256 eight-phase loop
        kern = gemm_select(x.dtype, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, rowstat=y)[0]
        nb = N // (32 if kern == GEMM_P8 else 64)
        st = torch.empty((M, nb, 2), dtype=torch.float32, device=x.device)
        # column sums of the stored logits per 64-row wave tile (the statistics epilogue of the 128 x 128 tile only)
        cst = torch.empty((M // 64, N), dtype=torch.float32, device=x.device) if (want_cs and kern != GEMM_P8) else None
        _gemm(x.dtype, A=x, B=w, C=y, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, rowstat=st, rowstat_center=cen, rowstat_scale=float(inv_temp) * LOG2E,
              colstat=cst, kernel=kern)
        mx = torch.empty((M,), dtype=torch.float32, device=x.device)
        lse = torch.empty_like(mx)
        check(lib.esvit_rowstat_combine(_p(st), M, nb, _p(mx), _p(lse), _stream()), "rowstat_combine")
        if cst is not None:
            mx.esvit_col_sums = colsum(cst)
        return y, mx, lse
    rows = M if out_rows is None else out_rows
    y = torch.empty((rows, N), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    pre = torch.empty((M, N), dtype=x.dtype, device=x.device) if (gelu and want_preact) else None
    _gemm(x.dtype, A=x, B=w, C=y, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias, residual=residual, ldr=N,
          rowmap=rowmap, rowmap_period=0 if rowmap is None else rowmap.numel(), rowmap_tokens=rowmap_tokens,
          rowscale=rowscale, rows_per_sample=rows_per_sample, aux=pre, ldaux=N,
          epilogue=(EPI_QGELU if quick else EPI_GELU) if gelu else EPI_NONE, out_f32=out_f32)
    return (y, pre) if gelu and want_preact else y


def linear_dgrad(dy, w, *, gelu_preact=None, out_f32=False, quick=False):
    """dx = dy @ w  (w [Nout, Kin] act, read k-strided); optional fused GELU': dx *= gelu'(preact) (quick: QuickGELU')."""
    dy, w = _actc(dy), _actc(w)
    M, Nout = dy.shape
    Kin = w.shape[1]
    assert w.shape[0] == Nout and dy.dtype == w.dtype
    dx = torch.empty((M, Kin), dtype=torch.float32 if out_f32 else dy.dtype, device=dy.device)
    if gelu_preact is None and Nout >= 4096:
        # a very long reduction (DINOHead last layer: K = out_dim) over few output tiles: the wide-tile loop, split-K to fill the chip
        kern = FORCE_GEMM_KERNEL
        if FORCE_GEMM_KERNEL == GEMM_AUTO and dy.dtype == torch.bfloat16 and Kin >= 192:
            # (the eight-phase loop addresses its operands with 32-bit DMA offsets: d(logits) below 4 GiB)
            kern = GEMM_P8 if (Nout % 64 == 0 and Kin % 256 == 0 and dy.numel() * 2 < 0xfff00000 and not os.environ.get("ESVIT_NO_P8_ROUTING")) else GEMM_DMA8
        _, tm, tn, slots = gemm_select(dy.dtype, M=M, N=Kin, K=Nout, lda=Nout, ldb=Kin, ldc=Kin, b_kstrided=1, kernel=kern)
        tiles = (-(-M // tm)) * (-(-Kin // tn))
        if tiles <= slots // 2:
            splitk = int(min(16, max(2, slots // tiles)))
            part = workspace(splitk * M * Kin, dy.device)
            _gemm(dy.dtype, A=dy, B=w, C=dx, M=M, N=Kin, K=Nout, lda=Nout, ldb=Kin, ldc=Kin, b_kstrided=1, out_f32=out_f32,
                  splitk=splitk, partial=part, kernel=kern)
            return dx
        _gemm(dy.dtype, A=dy, B=w, C=dx, M=M, N=Kin, K=Nout, lda=Nout, ldb=Kin, ldc=Kin, b_kstrided=1, out_f32=out_f32, kernel=kern)
        return dx
    _gemm(dy.dtype, A=dy, B=w, C=dx, M=M, N=Kin, K=Nout, lda=Nout, ldb=Kin, ldc=Kin, b_kstrided=1, aux=gelu_preact,
          ldaux=Kin, epilogue=(EPI_QGELU_BWD if quick else EPI_GELU_BWD) if gelu_preact is not None else EPI_NONE, out_f32=out_f32)
    return dx


def _pick_splitk(rows, Nout, Kin, tiles, slots=512):
    """split-K factor for wgrad: as many workgroups as fit the chip AT ONCE (`slots` resident workgroups: 256 CUs x 2
    64-KiB workgroups of the 4-wave loop, x 1 of the 8-wave loop) and never one more -- every workgroup runs for the whole
    kernel, so one extra doubles its duration --, >= 512 reduction rows per split, and partial slabs (written + re-read)
    no larger than the operand traffic"""
    want = max(1, slots // max(tiles, 1))
    by_rows = max(1, rows // 512)
    by_bytes = max(1, (rows * (Nout + Kin) * 2) // (Nout * Kin * 8))
    return int(max(1, min(want, by_rows, by_bytes, 512)))


def linear_wgrad(dy, x, *, out=None, accumulate=False, want_bias=False, db_out=None):
    """dw[Nout, Kin] = dy^T @ x (fp32), both operands read k-strided, split-K over the rows.
    want_bias: also return db[Nout] = column sums of dy, fused into the same kernel (all-ones MFMA fragment); db_out: the
    fp32 [Nout] tensor to write it to (e.g. the parameter's slot of a gradient bucket)."""
    dy, x = _actc(dy), _actc(x)
    rows, Nout = dy.shape
    Kin = x.shape[1]
    assert x.shape[0] == rows and dy.dtype == x.dtype
    if out is None:
        out = torch.empty((Nout, Kin), dtype=torch.float32, device=dy.device)
        accumulate = False
    db = None
    if want_bias:
        db = db_out if db_out is not None else torch.empty((Nout,), dtype=torch.float32, device=dy.device)
        assert db.shape == (Nout,) and db.dtype == torch.float32 and db.is_contiguous()
    _, tm, tn, slots = gemm_select(dy.dtype, M=Nout, N=Kin, K=rows, lda=Nout, ldb=Kin, ldc=Kin, a_kstrided=1, b_kstrided=1, colsum=db)
    tiles = (-(-Nout // tm)) * (-(-Kin // tn))
    splitk = _pick_splitk(rows, Nout, Kin, tiles, slots)
    if splitk > 1:
        part = workspace(splitk * Nout * (Kin + 1), dy.device)
        cpart = part[splitk * Nout * Kin:] if want_bias else None
        _gemm(dy.dtype, A=dy, B=x, C=out, M=Nout, N=Kin, K=rows, lda=Nout, ldb=Kin, ldc=Kin, a_kstrided=1, b_kstrided=1,
              out_f32=1, splitk=splitk, partial=part, accumulate=int(accumulate), colsum=db, colsum_partial=cpart)
    else:
        # accumulate through the residual input (each element is read and written by the same lane)
        _gemm(dy.dtype, A=dy, B=x, C=out, M=Nout, N=Kin, K=rows, lda=Nout, ldb=Kin, ldc=Kin, a_kstrided=1, b_kstrided=1,
              out_f32=1, residual=out if accumulate else None, ldr=Kin, colsum=db)
    return (out, db) if want_bias else out


def batched_nt(a, b, out_ld):
    """out[p] = a[p] @ b[p]^T in fp32: a [P, M, K], b [P, N, K] fp32 -> out [P, M, out_ld] (cols >= N undefined)."""
    a, b = _f32c(a), _f32c(b)
    P, M, K = a.shape
    N = b.shape[1]
    out = torch.empty((P, M, out_ld), dtype=torch.float32, device=a.device)
    _gemm(torch.float32, A=a, B=b, C=out, M=M, N=N, K=K, lda=K, ldb=K, ldc=out_ld, batch=P, strideA=M * K, strideB=N * K,
          strideC=M * out_ld, out_f32=1)
    return out


def colsum(x, *, out=None, accumulate=False):
    x = _actc(x)
    rows, N = x.shape
    if out is None:
        out = torch.empty((N,), dtype=torch.float32, device=x.device)
        accumulate = False
    ws = workspace(query(Q_COLSUM_BLOCKS, rows) * N, x.device, slot=1)
    check(lib.esvit_colsum(_code(x.dtype), _p(x), rows, N, N, _p(out), _p(ws), int(accumulate), _stream()), "colsum")
    return out


# ------------------------------------------------------------------------------------------------
# fused MLP forward
# ------------------------------------------------------------------------------------------------
def mlp_fused_supported(dt, Cc, backward=False):
    """the fused MLP forward exists for this width (inference passes); backward=True: the training pair exists"""
    return bool(query(Q_MLP_FUSED, _code(dt), int(Cc)) & (2 if backward else 1))


def mlp_fused_fwd(x, gamma, beta, eps, W1, b1, W2, b2, *, rowscale=None, next_norm=None):
    """x fp32 [M, C] -> y fp32 [M, C] = x + rowscale * (GELU(LN(x) W1^T + b1) W2^T + b2); nothing hidden-sized is written.
    next_norm = (gamma_next, beta_next): also return (xw_next act [M, C], mean_next, rstd_next) = LayerNorm(y) with the next
    block's norm1 parameters -> (y, (xw, mean, rstd))"""
    x, W1, W2 = _f32c(x), _actc(W1), _actc(W2)
    M, Cc = x.shape
    assert W1.shape == (4 * Cc, Cc) and W2.shape == (Cc, 4 * Cc) and W1.dtype == W2.dtype
    y = torch.empty_like(x)
    gn = bn = xw = mn = rn = None
    if next_norm is not None:
        gn, bn = _f32c(next_norm[0]), _f32c(next_norm[1])
        xw = torch.empty((M, Cc), dtype=W1.dtype, device=x.device)
        mn = torch.empty((M,), dtype=torch.float32, device=x.device)
        rn = torch.empty_like(mn)
    check(lib.esvit_mlp_fused_fwd(_code(W1.dtype), _p(x), _p(_f32c(gamma)), _p(_f32c(beta)), eps, _p(W1), _p(_f32c(b1)), _p(W2), _p(_f32c(b2)),
                                  _p(rowscale), M, Cc, _p(y), _p(gn), _p(bn), _p(xw), _p(mn), _p(rn), _stream()), "mlp_fused_fwd")
    return y if next_norm is None else (y, (xw, mn, rn))


def mlp_fused_fwd_train(x, gamma, beta, eps, W1, b1, W2, b2, *, rowscale=None):
    """the wide-stage (C = 384) training pass: y as mlp_fused_fwd plus what the UNFUSED backward reads, written by the same kernel:
    -> (y fp32 [M, C], a1 act [M, 4C] pre-activation, a1g act [M, 4C] GELU(a1), h act [M, C] LayerNorm(x), mean fp32 [M], rstd fp32 [M])"""
    x, W1, W2 = _f32c(x), _actc(W1), _actc(W2)
    M, Cc = x.shape
    assert W1.shape == (4 * Cc, Cc) and W2.shape == (Cc, 4 * Cc) and W1.dtype == W2.dtype
    y = torch.empty_like(x)
    a1 = torch.empty((M, 4 * Cc), dtype=W1.dtype, device=x.device)
    a1g = torch.empty_like(a1)
    h = torch.empty((M, Cc), dtype=W1.dtype, device=x.device)
    mean = torch.empty((M,), dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    check(lib.esvit_mlp_fused_fwd_train(_code(W1.dtype), _p(x), _p(_f32c(gamma)), _p(_f32c(beta)), eps, _p(W1), _p(_f32c(b1)), _p(W2), _p(_f32c(b2)),
                                        _p(rowscale), M, Cc, _p(y), _p(a1), _p(a1g), _p(h), _p(mean), _p(rstd), _stream()), "mlp_fused_fwd_train")
    return y, a1, a1g, h, mean, rstd


def mlp_fused_train_supported(dt, Cc, rows=0):
    """esvit_mlp_fused_fwd_train exists for this width (bf16, C = 384: forward fused, backward through the GEMMs) and row count (its side outputs
    are addressed through 32-bit offsets: [rows, 4C] bf16 below 2 GiB)"""
    return dt == torch.bfloat16 and int(Cc) == 384 and mlp_fused_supported(dt, Cc) and rows * 4 * Cc * 2 < 0x7ff00000


def cast_weight(w, transpose=False, perm32=False):
    """fp32 [R, S] -> bf16 ([S, R] if transpose), rows optionally in the 32-block channel order of the 16-token kernels"""
    w = _f32c(w)
    R, S = w.shape
    out = torch.empty((S, R) if transpose else (R, S), dtype=torch.bfloat16, device=w.device)
    check(lib.esvit_cast_weight(_p(w), _p(out), R, S, int(transpose), int(perm32), _stream()), "cast_weight")
    return out


MLP_W1_FWD, MLP_W1_BWD, MLP_W1T_BWD, MLP_W2T_BWD = range(4)  # ESVIT_MLP_* of include/esvit_hip.h


def mlp_fused_weight(kind, w):
    """the weight copy the fused MLP kernels stream, from the fp32 master: kind MLP_W1_FWD / MLP_W1_BWD (fc1.weight [4C, C] for
    esvit_mlp_fused_fwd / _bwd), MLP_W1T_BWD (its transpose), MLP_W2T_BWD (the transpose of fc2.weight [C, 4C]).  fc2.weight itself
    is used as the plain activation-dtype cast.  The channel order of a copy is private to the library."""
    w = _f32c(w)
    Cc = min(w.shape)
    out = torch.empty((w.shape[1], w.shape[0]) if kind in (MLP_W1T_BWD, MLP_W2T_BWD) else tuple(w.shape), dtype=torch.bfloat16, device=w.device)
    check(lib.esvit_mlp_fused_weight(int(kind), _p(w), _p(out), Cc, _stream()), "mlp_fused_weight")
    return out


def mlp_fused_bwd(x, gy, gamma, beta, eps, W1, W2T, W1T, b1, *, rowscale_mlp=None, rowscale_out=None):
    """data-gradient path of the fused MLP branch: x (branch input) fp32 [M, C], gy = dL/dy fp32 [M, C] ->
    (gx fp32 [M, C], gx_act act [M, C] = cast(rowscale_out * gx), xhat act [M, C], a1g act [M, 4C], da1 act [M, 4C])"""
    x, gy, W1, W2T, W1T = _f32c(x), _f32c(gy), _actc(W1), _actc(W2T), _actc(W1T)
    M, Cc = x.shape
    assert gy.shape == x.shape and W1.shape == (4 * Cc, Cc) and W2T.shape == (4 * Cc, Cc) and W1T.shape == (Cc, 4 * Cc)
    dt = W1.dtype
    gx = torch.empty_like(x)
    gxa = torch.empty((M, Cc), dtype=dt, device=x.device)
    xhat = torch.empty((M, Cc), dtype=dt, device=x.device)
    a1g = torch.empty((M, 4 * Cc), dtype=dt, device=x.device)
    da1 = torch.empty((M, 4 * Cc), dtype=dt, device=x.device)
    check(lib.esvit_mlp_fused_bwd(_code(dt), _p(x), _p(gy), _p(rowscale_mlp), _p(rowscale_out), _p(_f32c(gamma)), _p(_f32c(beta)), eps, _p(W1),
                                  _p(W2T), _p(W1T), _p(_f32c(b1)), M, Cc, _p(gx), _p(gxa), _p(xhat), _p(a1g), _p(da1), _stream()), "mlp_fused_bwd")
    return gx, gxa, xhat, a1g, da1


def ln_fold_finish(G, db, W, gamma, beta, *, gb_out=None):
    """LayerNorm folded out of a weight gradient: G = dY^T xhat [J, C] (overwritten by dW = G o gamma + db (x) beta), db = colsum(dY),
    W the fp32 master -> (dW, dgamma, dbeta)"""
    G, db, W = _f32c(G), _f32c(db), _f32c(W)
    J, Cc = G.shape
    assert W.shape == (J, Cc) and db.shape == (J,)
    dgamma, dbeta = _ln_grad_outs(Cc, G.device, gb_out)
    check(lib.esvit_ln_fold_finish(_p(G), _p(db), _p(W), _p(_f32c(gamma)), _p(_f32c(beta)), J, Cc, _p(dgamma), _p(dbeta), 0, _stream()), "ln_fold_finish")
    return G, dgamma, dbeta


# ------------------------------------------------------------------------------------------------
# normalisation
# ------------------------------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, eps, *, rowmap=None, period_out=0, out_rows=None, want_f32=False, dtype=None):
    """x fp32 [nB, T, C] or [rows, C] -> (y act, y_f32 or None, mean, rstd).  With rowmap (int32 [T], token ->
    window slot) y has out_rows rows, zero at the pad slots."""
    x = _f32c(x)
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    dt = dtype or _ACT_DTYPE
    tokens = 0
    if rowmap is not None:
        tokens = rowmap.numel()
        y = torch.zeros((out_rows, Cc), dtype=dt, device=x.device)
    else:
        y = torch.empty((rows, Cc), dtype=dt, device=x.device)
    yf = torch.empty((rows, Cc), dtype=torch.float32, device=x.device) if want_f32 else None
    mean = torch.empty((rows,), dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    check(lib.esvit_layernorm_fwd(_code(dt), _p(x), _p(gamma), _p(beta), eps, rows, Cc, _p(y), _p(yf), _p(mean), _p(rstd),
                                  _p(rowmap), tokens, period_out, _stream()), "layernorm_fwd")
    return y, yf, mean, rstd


def _ln_grad_outs(Cc, device, gb_out):
    """(dgamma, dbeta) destinations: the caller's pair of fp32 [C] tensors (gradient-bucket slots) or a fresh [2, C]"""
    if gb_out is not None and gb_out[0] is not None and gb_out[1] is not None:
        for t in gb_out:
            assert t.numel() == Cc and t.dtype == torch.float32 and t.is_contiguous()
        return gb_out[0].view(Cc), gb_out[1].view(Cc)
    gb = torch.empty((2, Cc), dtype=torch.float32, device=device)
    return gb[0], gb[1]


def layernorm_bwd(dy, x, mean, rstd, gamma, *, g_in=None, rowmap=None, period_in=0, gb_out=None):
    """-> (dx fp32 like x, dgamma, dbeta).  dy act, read through rowmap (token -> window slot) if given."""
    x, dy = _f32c(x), _actc(dy)
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    dx = torch.empty_like(x)
    dgamma, dbeta = _ln_grad_outs(Cc, x.device, gb_out)
    ws = workspace(query(Q_LN_BWD_BLOCKS, rows, Cc) * 2 * Cc, x.device, slot=1)
    check(lib.esvit_layernorm_bwd(_code(dy.dtype), _p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(g_in), rows, Cc, _p(dx),
                                  _p(dgamma), _p(dbeta), _p(ws), _p(rowmap), 0 if rowmap is None else rowmap.numel(),
                                  period_in, None, 0, None, 0, _stream()), "layernorm_bwd")
    return dx, dgamma, dbeta


def layernorm_bwd_cast(dy, x, mean, rstd, gamma, *, g_in=None, rowscale=None, rows_per_sample=0, gb_out=None):
    """-> (dx fp32, dx_act = cast(rowscale * dx) in dy's dtype, dgamma, dbeta): layernorm_bwd + gather_cast in one pass"""
    x, dy = _f32c(x), _actc(dy)
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    dx = torch.empty_like(x)
    dxa = torch.empty((rows, Cc), dtype=dy.dtype, device=x.device)
    dgamma, dbeta = _ln_grad_outs(Cc, x.device, gb_out)
    ws = workspace(query(Q_LN_BWD_BLOCKS, rows, Cc) * 2 * Cc, x.device, slot=1)
    check(lib.esvit_layernorm_bwd(_code(dy.dtype), _p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(g_in), rows, Cc, _p(dx), _p(dgamma),
                                  _p(dbeta), _p(ws), None, 0, 0, _p(dxa), _code(dxa.dtype), _p(rowscale), rows_per_sample, _stream()), "layernorm_bwd(cast)")
    return dx, dxa, dgamma, dbeta


def layernorm_bwd_to_act(dy, x, mean, rstd, gamma, *, gb_out=None):
    """-> (cast(dx) in the activation dtype, dgamma, dbeta) from an fp32 upstream gradient `dy`, read as it is: the LayerNorm of the patch
    embedding, whose dL/dy is the fp32 residual-stream gradient and whose dL/dx is only ever read as the operand of the projection's
    weight gradient -- no fp32 dx is written and no cast pass follows"""
    x, dy = _f32c(x), _f32c(dy)
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    dxa = torch.empty((rows, Cc), dtype=_ACT_DTYPE, device=x.device)
    dgamma, dbeta = _ln_grad_outs(Cc, x.device, gb_out)
    ws = workspace(query(Q_LN_BWD_BLOCKS, rows, Cc) * 2 * Cc, x.device, slot=1)
    check(lib.esvit_layernorm_bwd(_code(dy.dtype), _p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), None, rows, Cc, None, _p(dgamma),
                                  _p(dbeta), _p(ws), None, 0, 0, _p(dxa), _code(dxa.dtype), None, 0, _stream()), "layernorm_bwd(to act)")
    return dxa, dgamma, dbeta


def merge_ln_fwd(x, gamma, beta, eps, H, W, dtype=None, out=None):
    """x fp32 [nB, H*W, C] -> (y act [nB*H/2*W/2, 4C], mean, rstd).  out: optional preallocated (y, mean, rstd) -- e.g.
    row slices of buffers shared by several resolution groups."""
    x = _f32c(x)
    nB, L, Cc = x.shape
    assert L == H * W
    rows = nB * (H // 2) * (W // 2)
    dt = dtype or _ACT_DTYPE
    if out is not None:
        y, mean, rstd = out
        assert y.shape == (rows, 4 * Cc) and y.dtype == dt and y.is_contiguous() and mean.is_contiguous() and rstd.is_contiguous()
    else:
        y = torch.empty((rows, 4 * Cc), dtype=dt, device=x.device)
        mean = torch.empty((rows,), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
    check(lib.esvit_merge_ln_fwd(_code(dt), _p(x), _p(gamma), _p(beta), eps, nB, H, W, Cc, _p(y), _p(mean), _p(rstd), _stream()),
          "merge_ln_fwd")
    return y, mean, rstd


def merge_ln_bwd(dy, x, mean, rstd, gamma, H, W, dx_out=None, gb_out=None, accumulate=False, act_out=None, rowscale=None):
    """gb_out: optional fp32 [2, 4C] receiving (dgamma | dbeta); accumulate: add to it instead of overwriting (further
    resolution groups sharing the parameters).  act_out: optional activation-dtype [nB*H*W, C] receiving cast(rowscale * dx)
    (rowscale: per token row, or None) -- the MLP-branch operand of the block whose dL/dy this dx is (SwinBlockMultiFn's shadow)"""
    x, dy = _f32c(x), _actc(dy)
    nB, L, Cc = x.shape
    if dx_out is not None:
        assert dx_out.numel() == x.numel() and dx_out.dtype == torch.float32 and dx_out.is_contiguous()
        dx = dx_out
    else:
        dx = torch.empty_like(x)
    if act_out is not None:
        assert act_out.numel() == x.numel() and act_out.dtype == dy.dtype and act_out.is_contiguous()
        assert rowscale is None or (rowscale.numel() == nB * L and rowscale.dtype == torch.float32 and rowscale.is_contiguous())
    else:
        assert rowscale is None
    acc = bool(accumulate) and gb_out is not None
    gb = gb_out if gb_out is not None else torch.empty((2, 4 * Cc), dtype=torch.float32, device=x.device)
    assert gb.shape == (2, 4 * Cc) and gb.is_contiguous()
    dgamma, dbeta = gb[0], gb[1]
    rows = nB * (H // 2) * (W // 2)
    ws = workspace(query(Q_LN_BWD_BLOCKS, rows, 4 * Cc) * 8 * Cc, x.device, slot=1)
    check(lib.esvit_merge_ln_bwd(_code(dy.dtype), _p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), nB, H, W, Cc, _p(dx), _p(dgamma),
                                 _p(dbeta), _p(ws), _p(act_out), _p(rowscale), 1, int(acc), _stream()), "merge_ln_bwd")
    return dx, dgamma, dbeta


# ------------------------------------------------------------------------------------------------
# data movement
# ------------------------------------------------------------------------------------------------
def gather_cast(src, rows, *, rowmap=None, tokens=0, rowscale=None, rows_per_sample=0, dtype=None):
    """dst[r] = cast(scale * src[map(r)]); src fp32 [*, C] -> dst act [rows, C]."""
    src = _f32c(src)
    Cc = src.shape[-1]
    dt = dtype or _ACT_DTYPE
    dst = torch.empty((rows, Cc), dtype=dt, device=src.device)
    check(lib.esvit_gather_cast(_code(dt), _p(src), _p(dst), rows, Cc, _p(rowmap), 0 if rowmap is None else rowmap.numel(),
                                tokens, _p(rowscale), rows_per_sample, _stream()), "gather_cast")
    return dst


def cast_to_act(x, dtype=None, out=None):
    x = _f32c(x)
    dt = dtype or _ACT_DTYPE
    if out is None:
        out = torch.empty(x.shape, dtype=dt, device=x.device)
    assert out.dtype == dt and out.numel() == x.numel() and out.is_contiguous()
    check(lib.esvit_cast_f32_to(_code(dt), _p(x), _p(out), x.numel(), _stream()), "cast_f32_to")
    return out






def patch_im2col(img, P, Kpad, dtype=None, out=None):
    img = _f32c(img)
    nB, ch, S, S2 = img.shape
    assert ch == 3 and S == S2
    dt = dtype or _ACT_DTYPE
    G = S // P
    if out is not None:
        assert out.shape == (nB * G * G, Kpad) and out.dtype == dt and out.is_contiguous()
        cols = out
    else:
        cols = torch.empty((nB * G * G, Kpad), dtype=dt, device=img.device)
    check(lib.esvit_patch_im2col(_code(dt), _p(img), _p(cols), nB, S, P, Kpad, _stream()), "patch_im2col")
    return cols


def token_mean_fwd(x, dtype=None):
    """x fp32 [nB, T, C] -> (mean fp32 [nB, C], mean act [nB, C])."""
    x = _f32c(x)
    nB, T, Cc = x.shape
    dt = dtype or _ACT_DTYPE
    out = torch.empty((nB, Cc), dtype=torch.float32, device=x.device)
    out_act = torch.empty((nB, Cc), dtype=dt, device=x.device)
    check(lib.esvit_token_mean_fwd(_code(dt), _p(x), nB, T, Cc, _p(out), _p(out_act), _stream()), "token_mean_fwd")
    return out, out_act


def token_mean_bwd(g_mean, g_tok, T):
    g_mean = _f32c(g_mean)
    nB, Cc = g_mean.shape
    dx = torch.empty((nB, T, Cc), dtype=torch.float32, device=g_mean.device)
    check(lib.esvit_token_mean_bwd(_p(g_mean), _p(g_tok), nB, T, Cc, _p(dx), _stream()), "token_mean_bwd")
    return dx


# ------------------------------------------------------------------------------------------------
# window attention
# ------------------------------------------------------------------------------------------------
def attn_frag_elems(N):
    n = query(Q_ATTN_FRAG_ELEMS, N)
    if n < 0:
        raise RuntimeError("window size with %d tokens is not supported by the HIP attention kernel yet" % N)
    return n






def new_bias_frag(nH, N, device):
    """scratch for the fragment-order relative-position bias of one block (filled by the first attention call that gets it
    together with the table; later calls of the step -- the other resolution group, the backward -- pass it with table None)"""
    return torch.empty(2 * nH * attn_frag_elems(N), dtype=torch.float32, device=device)


def window_attn_fwd(qkv, qkv_bias, win2tok, L, rel_table, ws, region_ids, nW, N, nH, scale, want_attn=False, out=None, bias_frag=None):
    """token-ordered qkv [nB*L, 3C] -> (out [nB*L, C], lse or None[, attn]); win2tok int32 [nW*N] slot -> token (-1 = zero-pad slot).
    lse (per-query log-sum-exp) is produced for 14x14 windows, whose blocked backward needs it."""
    qkv = _actc(qkv)
    rows, C3 = qkv.shape
    Cc = C3 // 3
    nB = rows // L
    if out is None:
        out = torch.empty((rows, Cc), dtype=qkv.dtype, device=qkv.device)
    assert out.shape == (rows, Cc) and out.dtype == qkv.dtype and out.is_contiguous()  # may be a row slice of a larger matrix
    attn = torch.empty((nB * nW, nH, N, N), dtype=torch.float32, device=qkv.device) if want_attn else None
    nl = query(Q_ATTN_LSE_ELEMS, N)
    lse = torch.empty((nB * nW * nH, nl), dtype=torch.float32, device=qkv.device) if nl else None
    # frag-layout relative-position bias of every head: the caller's per-block buffer, or shared scratch refilled by this call
    bias_ws = bias_frag if bias_frag is not None else workspace(2 * nH * attn_frag_elems(N), qkv.device, slot=2)
    assert rel_table is not None or bias_frag is not None
    check(lib.esvit_window_attn_fwd(_code(qkv.dtype), _p(qkv), _p(_f32c(qkv_bias)), _p(win2tok), L, _p(None if rel_table is None else _f32c(rel_table)), ws, _p(bias_ws),
                                    _p(region_ids), nW, nB, N, nH, Cc // nH, scale, _p(out), _p(lse), _p(attn), _stream()),
          "window_attn_fwd")
    return (out, lse, attn) if want_attn else (out, lse)


def attn_branch_supported(dt, Cc, nH, N, rows=0, windows=0):
    """the fused attention branch (esvit_attn_branch_fwd) exists for this shape: bf16, head_dim 32, C in {96, 192}, <= 64-token windows,
    and -- when given -- a call of `rows` token rows / `windows` windows stays inside the kernel's 2 GiB buffer ranges and 2^22-window
    limit (esvit_hip.h), so that an oversized per-GPU batch takes the four-kernel sequence instead of raising"""
    return (dt == torch.bfloat16 and Cc in (96, 192) and Cc == 32 * nH and N <= 64
            and rows * 3 * Cc * 2 < 0x7fff0000 and rows * Cc * 4 < 0x7fff0000 and windows < (1 << 22))


def attn_branch_fwd(x, gamma, beta, eps, Wqkv_p, bqkv, Wproj_p, bproj, win2tok, L, rel_table, ws, region_ids, nW, N, nH, scale, *,
                    rowscale=None, out=None, bias_frag=None, save=False):
    """x fp32 [nB*L, C] token rows of one resolution group -> y fp32 = x + rowscale * (proj(window_attention(qkv(LayerNorm(x)))) + b_proj)
    in one kernel.  Wqkv_p / Wproj_p: cast_weight(W, perm32=True) of qkv.weight / proj.weight.  save=True also returns what the
    unfused backward reads: (y, (xw, mean, rstd, qkv, ao)); save=(xw, mean, rstd, qkv, ao) writes them into the given tensors (row
    slices of a stage's matrices).  bias_frag / rel_table as for window_attn_fwd."""
    x = _f32c(x)
    rows, Cc = x.shape
    nB = rows // L
    assert rows == nB * L and Wqkv_p.shape == (3 * Cc, Cc) and Wproj_p.shape == (Cc, Cc) and Wqkv_p.dtype == torch.bfloat16
    y = torch.empty_like(x) if out is None else out
    assert y.shape == x.shape and y.dtype == torch.float32 and y.is_contiguous()
    xw = qkv = ao = mean = rstd = None
    if save is True:
        xw = torch.empty((rows, Cc), dtype=torch.bfloat16, device=x.device)
        qkv = torch.empty((rows, 3 * Cc), dtype=torch.bfloat16, device=x.device)
        ao = torch.empty((rows, Cc), dtype=torch.bfloat16, device=x.device)
        mean = torch.empty((rows,), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
    elif save:
        xw, mean, rstd, qkv, ao = save
        assert xw.shape == (rows, Cc) and qkv.shape == (rows, 3 * Cc) and ao.shape == (rows, Cc) and mean.shape == (rows,) and rstd.shape == (rows,)
        assert all(t.is_contiguous() for t in save) and xw.dtype == qkv.dtype == ao.dtype == torch.bfloat16 and mean.dtype == rstd.dtype == torch.float32
    if rowscale is not None:
        assert rowscale.shape == (rows,) and rowscale.dtype == torch.float32 and rowscale.is_contiguous()
    bias_ws = bias_frag if bias_frag is not None else workspace(2 * nH * attn_frag_elems(N), x.device, slot=2)
    assert rel_table is not None or bias_frag is not None
    check(lib.esvit_attn_branch_fwd(BF16, _p(x), _p(_f32c(gamma)), _p(_f32c(beta)), eps, _p(Wqkv_p), _p(_f32c(bqkv)), _p(Wproj_p), _p(_f32c(bproj)),
                                    _p(win2tok), L, _p(None if rel_table is None else _f32c(rel_table)), ws, _p(bias_ws), _p(region_ids), nW, nB, N, nH,
                                    scale, _p(rowscale), _p(y), _p(xw), _p(qkv), _p(ao), _p(mean), _p(rstd), _stream()), "attn_branch_fwd")
    return (y, (xw, mean, rstd, qkv, ao)) if save else y


def attn_dbias_slabs(N, windows, nH, device):
    """One buffer for the bias-gradient slabs of several window_attn_bwd calls over the same window size (the resolution groups of a
    ragged block), so that ONE relpos_bias_bwd folds and scatters them all: -> (buffer [sum of parts, nH, frag], its per-call slices).
    windows: the calls' window counts (nB * nW)"""
    parts = [query(Q_ATTN_BWD_PARTS, N, w, nH) for w in windows]
    buf = torch.empty((sum(parts), nH, attn_frag_elems(N)), dtype=torch.float32, device=device)
    out, at = [], 0
    for p_ in parts:
        out.append(buf[at:at + p_])
        at += p_
    return buf, out


def window_attn_bwd(qkv, qkv_bias, win2tok, L, dout, fwd_out, lse, rel_table, ws, region_ids, nW, N, nH, scale, dqkv_out=None, bias_frag=None,
                    dbias_out=None):
    """-> (dqkv act [nB*L, 3C], dbias_ws fp32 [parts, nH, frag], dpad_ws fp32 [rows, 2C]).  dbias_out: this call's slice of attn_dbias_slabs"""
    qkv, dout = _actc(qkv), _actc(dout)
    rows, C3 = qkv.shape
    Cc = C3 // 3
    nB = rows // L
    code = _code(qkv.dtype)
    dqkv = torch.empty_like(qkv) if dqkv_out is None else dqkv_out
    assert dqkv.shape == qkv.shape and dqkv.dtype == qkv.dtype and dqkv.is_contiguous()
    parts = query(Q_ATTN_BWD_PARTS, N, nB * nW, nH)
    if dbias_out is None:
        dbias_ws = torch.empty((parts, nH, attn_frag_elems(N)), dtype=torch.float32, device=qkv.device)
    else:
        dbias_ws = _f32c(dbias_out)
        assert tuple(dbias_ws.shape) == (parts, nH, attn_frag_elems(N)), (dbias_ws.shape, parts, nH)
    # the <= 64-token kernel writes every element of its pad-row slab itself; the 14x14 kernels fill one head's slice per row
    alloc = torch.empty if N <= 64 else torch.zeros
    pad = alloc((query(Q_ATTN_BWD_PAD_ROWS, N, nB * nW, nH | (code << 32)), 2 * Cc), dtype=torch.float32, device=qkv.device)
    bias_ws = bias_frag if bias_frag is not None else workspace(2 * nH * attn_frag_elems(N), qkv.device, slot=2)
    assert rel_table is not None or bias_frag is not None
    check(lib.esvit_window_attn_bwd(code, _p(qkv), _p(_f32c(qkv_bias)), _p(win2tok), L, _p(dout), _p(fwd_out), _p(lse), _p(None if rel_table is None else _f32c(rel_table)),
                                    ws, _p(bias_ws), _p(region_ids), nW, nB, N, nH, Cc // nH, scale, _p(dqkv), _p(dbias_ws), _p(pad),
                                    _stream()), "window_attn_bwd")
    return dqkv, dbias_ws, pad


def relpos_bias_bwd(dbias_ws, index, N, table_rows, out=None, accumulate=None):
    """out: optional fp32 [table_rows, nH]; accumulate (default: True when out is given -- the second resolution group of a
    ragged block) adds to it, False overwrites it (a gradient-bucket slot written for the first time)"""
    parts, nH, _ = dbias_ws.shape
    dtable = torch.empty((table_rows, nH), dtype=torch.float32, device=dbias_ws.device) if out is None else _f32c(out)
    acc = (out is not None) if accumulate is None else (bool(accumulate) and out is not None)
    check(lib.esvit_relpos_bias_bwd(_p(dbias_ws), parts, _p(index), N, nH, table_rows, _p(dtable), int(acc), _stream()),
          "relpos_bias_bwd")
    return dtable


# ------------------------------------------------------------------------------------------------
# DINOHead pieces
# ------------------------------------------------------------------------------------------------
def l2norm_fwd(x):
    x = _actc(x)
    R, D = x.shape
    z = torch.empty_like(x)
    inv = torch.empty((R,), dtype=torch.float32, device=x.device)
    check(lib.esvit_l2norm_fwd(_code(x.dtype), _p(x), R, D, _p(z), _p(inv), _stream()), "l2norm_fwd")
    return z, inv


def l2norm_bwd(dz, z, inv):
    dz, z = _actc(dz), _actc(z)
    R, D = z.shape
    dx = torch.empty_like(z)
    check(lib.esvit_l2norm_bwd(_code(z.dtype), _p(dz), _p(z), _p(inv), R, D, _p(dx), _stream()), "l2norm_bwd")
    return dx


def weightnorm_fwd(v, g, dtype=None):
    """v fp32 [K, D], g fp32 [K, 1] -> (w act [K, D], inv_norm fp32 [K])."""
    v, g = _f32c(v), _f32c(g)
    K, D = v.shape
    dt = dtype or _ACT_DTYPE
    w = torch.empty((K, D), dtype=dt, device=v.device)
    inv = torch.empty((K,), dtype=torch.float32, device=v.device)
    check(lib.esvit_weightnorm_fwd(_code(dt), _p(v), _p(g), K, D, _p(w), None, _p(inv), _stream()), "weightnorm_fwd")
    return w, inv


def weightnorm_bwd(dw, v, g, inv, need_dg, dv_out=None):
    dw, v, g = _f32c(dw), _f32c(v), _f32c(g)
    K, D = v.shape
    dv = torch.empty_like(v) if dv_out is None else _f32c(dv_out)
    assert dv.shape == v.shape
    dg = torch.empty((K, 1), dtype=torch.float32, device=v.device) if need_dg else None
    check(lib.esvit_weightnorm_bwd(_p(dw), _p(v), _p(g), _p(inv), K, D, _p(dv), _p(dg), _stream()), "weightnorm_bwd")
    return dv, dg


# ------------------------------------------------------------------------------------------------
# loss
# ------------------------------------------------------------------------------------------------
def teacher_row_stats(t, center, inv_temp):
    t = _actc(t)
    R, K = t.shape
    mx = torch.empty((R,), dtype=torch.float32, device=t.device)
    lse = torch.empty_like(mx)
    check(lib.esvit_teacher_row_stats(_code(t.dtype), _p(t), _p(center), inv_temp, R, K, _p(mx), _p(lse), _stream()),
          "teacher_row_stats")
    return mx, lse




def region_match(sim, Tt, crop_id, cm_row, tmatch):
    """sim fp32 [B, S, ld]; writes tmatch (int32 [rows, 2]) in place."""
    sim = _f32c(sim)
    B, S, ld = sim.shape
    check(lib.esvit_region_match(_p(sim), B, S, Tt, ld, _p(crop_id), _p(cm_row), _p(tmatch), _stream()), "region_match")
    return tmatch


def dino_ce(s, t, center, t_max, t_lse, tmatch, row_w, inv_student_temp, inv_teacher_temp, row_loss=None, term_w=None, row_order=None,
            s_stats=None):
    """-> (row_loss fp32 [Rs], ds act [Rs, K]).  tmatch int32 [Rs, 2] with one weight row_w[r] for both terms, or (mixup
    targets) tmatch [Rs, 4] with term_w fp32 [Rs, 4], one weight per term.  s_stats = (row_max, row_lse) of s * inv_student_temp
    (linear_fwd(row_stats=(inv_student_temp, None))) spares the kernel its first pass over the student rows."""
    s, t = _actc(s), _actc(t)
    Rs, K = s.shape
    assert t.shape[1] == K and s.dtype == t.dtype and tmatch.dtype == torch.int32
    if row_loss is None:
        row_loss = torch.empty((Rs,), dtype=torch.float32, device=s.device)
    assert row_loss.numel() == Rs and row_loss.is_contiguous()
    ds = torch.empty_like(s)
    terms = 2 if term_w is None else 4
    assert row_order is None or (row_order.dtype == torch.int32 and row_order.numel() == Rs and row_order.is_contiguous())
    assert tmatch.numel() == Rs * terms and tmatch.is_contiguous() and (term_w is None or (term_w.numel() == Rs * 4 and term_w.is_contiguous()))
    s_mx, s_lse = (None, None) if s_stats is None else s_stats
    assert s_stats is None or (term_w is None and s_mx.numel() == Rs and s_lse.numel() == Rs and s_mx.dtype == s_lse.dtype == torch.float32)
    check(lib.esvit_dino_ce_fwd_bwd(_code(s.dtype), _p(s), _p(t), _p(center), _p(t_max), _p(t_lse), _p(tmatch), _p(row_w), terms,
                                    _p(term_w), inv_student_temp, inv_teacher_temp, Rs, K, _p(row_loss), _p(ds), _p(row_order), _p(s_mx),
                                    _p(s_lse), _stream()),
          "dino_ce_fwd_bwd")
    return row_loss, ds


def sum_f32(x):
    x = _f32c(x)
    out = torch.empty((), dtype=torch.float32, device=x.device)
    check(lib.esvit_sum_f32(_p(x), x.numel(), _p(out), _stream()), "sum_f32")
    return out


def scale_inplace(x, scale):
    x = _actc(x)
    check(lib.esvit_scale_inplace(_code(x.dtype), _p(x), x.numel(), _p(_f32c(scale)), _stream()), "scale_inplace")
    return x


def center_ema(center, colsum_, momentum, denom):
    check(lib.esvit_center_ema(_p(_f32c(center)), _p(_f32c(colsum_)), momentum, float(denom), center.numel(), _stream()),
          "center_ema")
    return center


# ------------------------------------------------------------------------------------------------
# fused update
# ------------------------------------------------------------------------------------------------
def update_chunk_elems():
    return query(Q_UPDATE_CHUNK_ELEMS)


RULE_ADAMW, RULE_SGD, RULE_LARS = 0, 1, 2


def grad_sqnorm(tensors, ntensors, chunks, nchunks, sqnorms, stats=1):
    check(lib.esvit_grad_sqnorm(_p(tensors), ntensors, _p(chunks), nchunks, int(stats), _p(sqnorms), _stream()), "grad_sqnorm")


def fused_clip_update_ema(rule, tensors, ntensors, chunks, nchunks, sqnorms, clip, lr, wd, beta1, beta2, eps, ema_m, skipped=None):
    """rule AdamW: (beta1, beta2, eps); SGD: beta1 = momentum; LARS: beta1 = momentum, beta2 = eta"""
    check(lib.esvit_fused_clip_update_ema(int(rule), _p(tensors), ntensors, _p(chunks), nchunks, _p(sqnorms), clip, lr, wd, beta1, beta2,
                                          eps, ema_m, _p(skipped), _stream()), "fused_clip_update_ema")


# ------------------------------------------------------------------------------------------------
# CvT backbone pieces (cvt_v4_transformer.py): ConvEmbed im2col, depthwise 3x3, BatchNorm reductions
# ------------------------------------------------------------------------------------------------
def conv_out_size(n, k, stride, pad):
    return (n + 2 * pad - k) // stride + 1


def conv_im2col(src, nchw, nB, H, W, Cin, k, stride, pad, dtype=None):
    """-> cols [nB*Ho*Wo, Kpad] (Kpad = k*k*Cin rounded up to 8), column order (ky, kx, c).
    src: fp32 NCHW images (nchw=True) or activation-dtype NHWC tokens [nB*H*W, Cin]."""
    dt = dtype or (_ACT_DTYPE if nchw else src.dtype)
    Ho, Wo = conv_out_size(H, k, stride, pad), conv_out_size(W, k, stride, pad)
    Kpad = -(-(k * k * Cin) // 8) * 8
    assert src.is_contiguous() and (src.dtype == torch.float32 if nchw else src.dtype == dt)
    cols = torch.empty((nB * Ho * Wo, Kpad), dtype=dt, device=src.device)
    check(lib.esvit_conv_im2col(_code(dt), _p(src), int(bool(nchw)), nB, H, W, Cin, k, stride, pad, Ho, Wo, Kpad, _p(cols), _stream()),
          "conv_im2col")
    return cols


def conv_col2im(dcols, nB, H, W, Cin, k, stride, pad):
    """adjoint of conv_im2col for NHWC sources: dcols act [nB*Ho*Wo, Kpad] -> dsrc fp32 [nB*H*W, Cin]."""
    dcols = _actc(dcols)
    Ho, Wo = conv_out_size(H, k, stride, pad), conv_out_size(W, k, stride, pad)
    dsrc = torch.empty((nB * H * W, Cin), dtype=torch.float32, device=dcols.device)
    check(lib.esvit_conv_col2im(_code(dcols.dtype), _p(dcols), nB, H, W, Cin, k, stride, pad, Ho, Wo, dcols.shape[1], _p(dsrc), _stream()),
          "conv_col2im")
    return dsrc


def dwconv3x3(x, w, nB, H, W, flip=False):
    """depthwise 3x3, stride 1, zero pad 1 on NHWC tokens x [nB*H*W, C] (act); w fp32 [C, 9]; flip: mirrored taps."""
    x = _actc(x)
    Cc = x.shape[1]
    assert x.shape[0] == nB * H * W and w.numel() == 9 * Cc
    y = torch.empty_like(x)
    check(lib.esvit_dwconv3x3(_code(x.dtype), _p(x), _p(_f32c(w)), int(bool(flip)), nB, H, W, Cc, _p(y), _stream()), "dwconv3x3")
    return y


def dwconv3x3_wgrad(x, dy, nB, H, W):
    """-> dw fp32 [C, 9]"""
    x, dy = _actc(x), _actc(dy)
    Cc = x.shape[1]
    assert x.dtype == dy.dtype and x.shape == dy.shape
    dw = torch.empty((Cc, 9), dtype=torch.float32, device=x.device)
    ws = workspace(query(Q_COL_REDUCE_BLOCKS, nB * H * W) * 9 * Cc, x.device, slot=1)
    check(lib.esvit_dwconv3x3_wgrad(_code(x.dtype), _p(x), _p(dy), nB, H, W, Cc, _p(dw), _p(ws), _stream()), "dwconv3x3_wgrad")
    return dw


def col_sums2(a, b):
    """-> fp32 [2, C]: sum_r a[r, c]  and  sum_r a[r, c] * b[r, c]"""
    a, b = _actc(a), _actc(b)
    rows, Cc = a.shape
    assert a.dtype == b.dtype and a.shape == b.shape
    out = torch.empty((2, Cc), dtype=torch.float32, device=a.device)
    ws = workspace(query(Q_COL_REDUCE_BLOCKS, rows) * 2 * Cc, a.device, slot=1)
    check(lib.esvit_col_sums2(_code(a.dtype), _p(a), _p(b), rows, Cc, _p(out), _p(ws), _stream()), "col_sums2")
    return out


def col_affine2(x1, a1, a3, x2=None, a2=None, act=0):
    """act 0: y = a1[c] * x1 + a2[c] * x2 + a3[c];  1: y = GELU(a1[c] * x1 + a3[c]);  2: y = x2 * GELU'(a1[c] * x1 + a3[c]);
    3: y = max(a1[c] * x1 + a3[c], 0);  4: y = x2 where a1[c] * x1 + a3[c] > 0 else 0  (act dtype in / out, fp32 coefficients)"""
    x1 = _actc(x1)
    rows, Cc = x1.shape
    y = torch.empty_like(x1)
    check(lib.esvit_col_affine2(_code(x1.dtype), _p(x1), _p(x2), rows, Cc, _p(_f32c(a1)), _p(a2), _p(_f32c(a3)), int(act), _p(y), _stream()),
          "col_affine2")
    return y


def bn_fwd_coeffs(sums, n, gamma, beta, eps, momentum, running_mean=None, running_var=None):
    """sums fp32 [2, C] (sum d, sum d^2; already summed over the ranks) -> coef fp32 [4, C] = (a, shift, mean, rstd)"""
    Cc = gamma.numel()
    coef = torch.empty((4, Cc), dtype=torch.float32, device=sums.device)
    check(lib.esvit_bn_fwd_coeffs(_p(_f32c(sums)), float(n), _p(_f32c(gamma)), _p(_f32c(beta)), float(eps), float(momentum),
                                  _p(running_mean), _p(running_var), Cc, _p(coef), _stream()), "bn_fwd_coeffs")
    return coef


def bn_eval_coeffs(running_mean, running_var, gamma, beta, eps):
    Cc = gamma.numel()
    coef = torch.empty((4, Cc), dtype=torch.float32, device=gamma.device)
    check(lib.esvit_bn_eval_coeffs(_p(_f32c(running_mean)), _p(_f32c(running_var)), _p(_f32c(gamma)), _p(_f32c(beta)), float(eps), Cc,
                                   _p(coef), _stream()), "bn_eval_coeffs")
    return coef


def bn_bwd_local(sums, coef):
    """sums fp32 [2, C] (sum dy, sum dy*d) -> red fp32 [2, C] (sum dy, sum dy*xhat)"""
    Cc = coef.shape[1]
    red = torch.empty((2, Cc), dtype=torch.float32, device=sums.device)
    check(lib.esvit_bn_bwd_local(_p(_f32c(sums)), _p(_f32c(coef)), Cc, _p(red), _stream()), "bn_bwd_local")
    return red


def bn_bwd_coeffs(red, n, gamma, coef):
    """-> abc fp32 [3, C]: d(d) = A*dy + B*d + C (red=None: fixed / eval statistics)"""
    Cc = coef.shape[1]
    abc = torch.empty((3, Cc), dtype=torch.float32, device=coef.device)
    check(lib.esvit_bn_bwd_coeffs(_p(red), float(n), _p(_f32c(gamma)), _p(_f32c(coef)), Cc, _p(abc), _stream()), "bn_bwd_coeffs")
    return abc


def pad_crop_tokens(src, nB, Hs, Ws, Hd, Wd):
    """[nB*Hs*Ws, C] -> [nB*Hd*Wd, C]: zero-pad at the bottom / right, or crop"""
    src = _actc(src)
    Cc = src.shape[1]
    dst = torch.empty((nB * Hd * Wd, Cc), dtype=src.dtype, device=src.device)
    check(lib.esvit_pad_crop_tokens(_code(src.dtype), _p(src), nB, Hs, Ws, Hd, Wd, Cc, _p(dst), _stream()), "pad_crop_tokens")
    return dst


# ------------------------------------------------------------------------------------------------
# crop producer (datasets/build.py:203-261)
# ------------------------------------------------------------------------------------------------
AUG_PARAM_INTS = 24


def aug_max_box(S):
    """largest crop-box side aug_crops can resize to S x S"""
    return query(Q_AUG_MAX_BOX, S)


def aug_crops(src, images, params, S, max_h, max_w, planes=None, out=None):
    """DataAugmentationDINO for the n crops of one output size S described by ``params`` (int32 [n, 24], include/esvit_hip.h):
    src uint8 packed HWC images (readable up to the next 4-byte boundary), images int64 [n_img, 3] (byte offset, H, W).
    ``planes``: uint8 scratch of n * (3 S^2 + 4) bytes.  Returns (out fp32 [n, 3, S, S], uint8 [n, 3, S, S] view of the scratch =
    the resized, flipped crops before the jitter)."""
    assert src.is_cuda and src.dtype == torch.uint8 and src.is_contiguous()
    assert images.is_cuda and images.dtype == torch.int64 and images.is_contiguous() and images.shape[-1] == 3
    assert params.is_cuda and params.dtype == torch.int32 and params.is_contiguous() and params.shape[-1] == AUG_PARAM_INTS
    n = params.shape[0]
    need = n * (3 * S * S + 4)
    if planes is None:
        planes = torch.empty(need, dtype=torch.uint8, device=src.device)
    if out is None:
        out = torch.empty((n, 3, S, S), dtype=torch.float32, device=src.device)
    assert planes.is_cuda and planes.dtype == torch.uint8 and planes.is_contiguous() and planes.numel() >= need
    assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.numel() >= n * 3 * S * S
    check(lib.esvit_aug_crops(_p(src), _p(images), _p(params), n, S, int(max_h), int(max_w), _p(planes), _p(out), _stream()), "aug_crops")
    return out, planes.view(-1)[:n * 3 * S * S].view(n, 3, S, S)


# ------------------------------------------------------------------------------------------------
# global attention of the monolithic ViT backbones (models/vision_transformer.py:67-94)
# ------------------------------------------------------------------------------------------------
def vit_pad_tokens(N):
    """tokens per image rounded up to the granularity the batched GEMMs need (k-strided operands: multiples of 16)"""
    return -(-N // 16) * 16


def heads_split(x, B, N, nH, parts):
    """x [B * N, parts * C] token-major -> [parts, B, nH, Np, hd] (pad rows zero)"""
    x = _actc(x)
    C_ = x.shape[1] // parts
    hd, Np = C_ // nH, vit_pad_tokens(N)
    y = torch.empty((parts, B, nH, Np, hd), dtype=x.dtype, device=x.device)
    check(lib.esvit_heads_split(_code(x.dtype), _p(x), B, N, Np, nH, hd, parts, _p(y), _stream()), "heads_split")
    return y


def heads_merge(y, N):
    """y [parts, B, nH, Np, hd] -> [B * N, parts * nH * hd] token-major"""
    y = _actc(y)
    parts, B, nH, Np, hd = y.shape
    x = torch.empty((B * N, parts * nH * hd), dtype=y.dtype, device=y.device)
    check(lib.esvit_heads_merge(_code(y.dtype), _p(y), B, N, Np, nH, hd, parts, _p(x), _stream()), "heads_merge")
    return x


def _bmm(a, b, M, N, K, *, a_kstrided=0, b_kstrided=0, out=None):
    """batched esvit_gemm over the leading axis of contiguous 3-D operands: out[z] = op(a[z]) @ op(b[z]) -> [Z, M, N]"""
    Z = a.shape[0]
    if out is None:
        out = torch.empty((Z, M, N), dtype=a.dtype, device=a.device)
    assert out.is_contiguous() and out.shape == (Z, M, N) and a.is_contiguous() and b.is_contiguous()
    _gemm(a.dtype, A=a, B=b, C=out, M=M, N=N, K=K, lda=a.shape[2], ldb=b.shape[2], ldc=N, a_kstrided=a_kstrided, b_kstrided=b_kstrided,
          batch=Z, strideA=a.shape[1] * a.shape[2], strideB=b.shape[1] * b.shape[2], strideC=M * N)
    return out


def vit_attn_fwd(qkv, B, N, nH, scale, chunk=None):
    """Attention.forward between the qkv and proj projections (vision_transformer.py:76-83): qkv [B * N, 3C] -> (out [B * N, C],
    saved = (q | k | v [3, B nH, Np, hd], P [B nH, Np, Np])).  chunk (int32 [N]): Vision Longformer's sliding-chunk neighbourhood
    (esvit_softmax_rows_chunked_fwd) instead of global attention; or (table, nglo, tokens per chunk row) when the local tokens are
    ordered chunk row by chunk row, which lets the kernels skip the chunk rows a query cannot see."""
    C_ = qkv.shape[1] // 3
    hd, Np = C_ // nH, vit_pad_tokens(N)
    qkvh = heads_split(qkv, B, N, nH, 3).view(3, B * nH, Np, hd)
    prob = _bmm(qkvh[0], qkvh[1], Np, Np, hd)                                  # S = q k^T
    if chunk is not None:
        tab, nglo, rowtok = chunk if isinstance(chunk, tuple) else (chunk, 0, 0)
        assert tab.dtype == torch.int32 and tab.numel() == N and tab.is_contiguous()
        check(lib.esvit_softmax_rows_chunked_fwd(_code(prob.dtype), _p(prob), B * nH, N, Np, float(scale), _p(tab), int(nglo), int(rowtok), _stream()),
              "softmax_rows_chunked_fwd")
    else:
        check(lib.esvit_softmax_rows_fwd(_code(prob.dtype), _p(prob), B * nH, N, Np, float(scale), _stream()), "softmax_rows_fwd")
    o = _bmm(prob, qkvh[2], Np, hd, Np, b_kstrided=1)                          # O = P v
    return heads_merge(o.view(1, B, nH, Np, hd), N), (qkvh, prob)


def vit_attn_bwd(dout, saved, B, N, nH, scale, chunk=None):
    """gradient of vit_attn_fwd with respect to qkv: dout [B * N, C] -> dqkv [B * N, 3C] (chunk: as given to the forward; only the
    (table, nglo, tokens per chunk row) form changes anything -- the backward then skips the columns that are zero in P)"""
    qkvh, prob = saved
    _, Z, Np, hd = qkvh.shape
    do = heads_split(dout, B, N, nH, 1).view(Z, Np, hd)
    dqkvh = torch.empty_like(qkvh)
    _bmm(prob, do, Np, hd, Np, a_kstrided=1, b_kstrided=1, out=dqkvh[2])      # dv = P^T dO
    dp = _bmm(do, qkvh[2], Np, Np, hd)                                         # dP = dO v^T
    if isinstance(chunk, tuple) and chunk[2] > 0:
        check(lib.esvit_softmax_rows_chunked_bwd(_code(dp.dtype), _p(prob), _p(dp), Z, N, Np, float(scale), _p(chunk[0]), int(chunk[1]), int(chunk[2]),
                                                 _stream()), "softmax_rows_chunked_bwd")
    else:
        check(lib.esvit_softmax_rows_bwd(_code(dp.dtype), _p(prob), _p(dp), Z, N, Np, float(scale), _stream()), "softmax_rows_bwd")
    _bmm(dp, qkvh[1], Np, hd, Np, b_kstrided=1, out=dqkvh[0])                  # dq = dS k
    _bmm(dp, qkvh[0], Np, hd, Np, a_kstrided=1, b_kstrided=1, out=dqkvh[1])   # dk = dS^T q
    return heads_merge(dqkvh.view(3, B, nH, Np, hd), N)
