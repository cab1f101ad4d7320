"""Autograd glue: each Function composes the C-ABI kernels (esvit_amd.ops) for one stage of the hot
path and carries the hand-derived backward.  No torch compute op is used on
the data path -- torch provides allocation, streams and the autograd graph only.

Stages (reference lines):
  SwinBlockFn      swin_transformer.py:275-333 (+120-152): LN -> pad/roll/partition -> qkv -> window
                   attention -> proj -> reverse/unroll/crop + residual -> LN -> MLP + residual
  PatchEmbedFn     swin_transformer.py:537-547
  PatchMergeFn     swin_transformer.py:393-420
  FinalNormFn      swin_transformer.py:687
  TokenMeanFn      swin_transformer.py:688-689
  DinoHeadFn       vision_transformer.py:414-418
  ConvEmbedFn, CvtAttnFn, CvtFfnFn   cvt_v4_transformer.py:349-382, 108-220 (+75-105), 62-72  (BASELINE config 5)
"""
import os

import torch

from . import ops
from . import params as P

LN_EPS = 1e-6


def ops_module():
    return ops


# ------------------------------------------------------------------------------------------------
# static per-geometry tables (index maps, shift masks), cached per device
# ------------------------------------------------------------------------------------------------
_GEOM = {}


class WindowGeometry:
    """Everything that depends only on (H, W, window, shift): the bit-exact slot<->token maps and the
    shift mask in the attention kernel's fragment layout."""

    def __init__(self, H, W, ws, shift, device):
        o = ops_module()
        win2tok, tok2win = o.window_maps(H, W, ws, shift)
        self.H, self.W, self.ws, self.shift = H, W, ws, shift
        self.N = ws * ws
        self.period = int(win2tok.size)           # window slots per image
        self.nW = self.period // self.N
        self.tokens = H * W
        self.win2tok = torch.from_numpy(win2tok).to(device)
        self.tok2win = torch.from_numpy(tok2win).to(device)
        self.region_ids = None
        if shift > 0:
            self.region_ids = torch.from_numpy(o.shift_region_ids(H, W, ws, shift)).to(device)


def geometry(H, W, ws, shift, device):
    key = (H, W, ws, shift, str(device), id(ops_module()))
    g = _GEOM.get(key)
    if g is None:
        g = WindowGeometry(H, W, ws, shift, device)
        _GEOM[key] = g
    return g


def _alias(t, sink):
    """a gradient that was written into a bucket slot goes to autograd as a FRESH alias of the slot (AccumulateGrad only adopts
    a tensor nobody else holds; it would clone the stored view)"""
    return t.detach() if sink is not None else t


def _wgrad(dy, x, param, shape2d=None, want_bias=False, bias_param=None):
    """weight (and bias) gradient of `param` = dy^T x; written straight into the data-parallel reducer's bucket slots when
    they are armed (params.grad_out), and handed to autograd as fresh aliases so that AccumulateGrad adopts them without a copy"""
    o = ops_module()
    sink = P.grad_out(param, shape2d)
    if not want_bias:
        return _alias(o.linear_wgrad(dy, x, out=sink), sink)
    sink_b = P.grad_out(bias_param) if bias_param is not None else None
    dw, db = o.linear_wgrad(dy, x, out=sink, want_bias=True, db_out=sink_b)
    return _alias(dw, sink), _alias(db, sink_b)



# ---- weight gradients on a second stream -------------------------------------------------------------------------------------------
# The weight-gradient GEMMs are leaves of the backward's dependency chain: nothing downstream in the same node reads them.  A Swin
# block's / head's / PatchMerging's backward launches them on a side stream behind an event and joins before it returns --
# every gradient is complete on the autograd stream when the node hands it over -- so that they share the CUs with the dgrad /
# LayerNorm / attention kernels of the chain (same-box A-B, profiles/r06_wgrad_stream_ab.txt: 48.47 -> 47.96 ms per step).  The stream has
# the default priority: a high-priority one measured 0.09 ms better alone and 10 ms WORSE beside RCCL's stream (ibid.).
# ESVIT_WGRAD_STREAM=0 puts them back in line.
WGRAD_STREAM = os.environ.get("ESVIT_WGRAD_STREAM", "1") != "0"
_wg_state = {}


def _side_run(fn, *keep):
    """fn() launches kernels whose inputs are ready on the current stream; `keep`: the tensors they read (held until _side_join, so
    that the caching allocator cannot hand their memory to the autograd stream while the side stream still reads it)"""
    if not (WGRAD_STREAM and keep[0].is_cuda):
        return fn()
    dev = keep[0].device.index
    st = _wg_state.get(dev)
    if st is None:
        st = _wg_state[dev] = {"stream": torch.cuda.Stream(device=dev, priority=int(os.environ.get("ESVIT_WGRAD_PRIO", "0"))), "keep": [],
                               "pending": False, "event": torch.cuda.Event()}
    ev = st["event"]  # one event per device, re-recorded: a wait refers to the record that precedes it in program order
    ev.record(torch.cuda.current_stream(dev))
    st["stream"].wait_event(ev)
    with torch.cuda.stream(st["stream"]):
        out = fn()
    st["keep"].append(keep)
    st["pending"] = True
    return out


_DEFER_PROBE = os.environ.get("ESVIT_PROBE_WGRAD_DEFER", "0") == "1"  # timing probe only (UNSAFE: gradients may be read before they are complete)


def _side_join(final=False):
    if _DEFER_PROBE and not final:
        return
    for dev, st in _wg_state.items():
        if st["pending"]:
            torch.cuda.current_stream(dev).wait_stream(st["stream"])
            st["keep"].clear()
            st["pending"] = False


def _ln_sinks(gparam, bparam):
    """bucket slots of a LayerNorm's (weight, bias) gradients, or None when no reducer is armed / either has none"""
    sg, sb = P.grad_out(gparam), P.grad_out(bparam)
    return (sg, sb) if sg is not None and sb is not None else None


def _weight(p, shape2d=None):
    """activation-dtype copy of an fp32 parameter; frozen parameters (the EMA teacher) are recast on every
    use because in-place `.data` updates (main_esvit.py:590) are invisible to version counters."""
    if p.requires_grad or P.is_managed(p):
        return P.cached_cast(p, shape2d)
    src = p.detach() if shape2d is None else p.detach().reshape(shape2d)
    return ops_module().cast_to_act(src.contiguous())


def _mlp_w(p, kind):
    """the copy of fc1.weight / fc2.weight the fused MLP kernels stream (ops.mlp_fused_weight), cached per parameter version like
    the plain cast; frozen parameters that nobody manages (a teacher outside the fused updater) are converted on every use"""
    o = ops_module()
    fn = lambda: o.mlp_fused_weight(getattr(o, kind), p.detach().contiguous())  # noqa: E731
    if p.requires_grad or P.is_managed(p):
        return P.cached(p, kind, fn)
    return fn()


def _perm_w(p):
    """the copy of qkv.weight / proj.weight the fused attention branch streams (ops.cast_weight, perm32), cached like _mlp_w"""
    o = ops_module()
    fn = lambda: o.cast_weight(p.detach().contiguous(), perm32=True)  # noqa: E731
    if p.requires_grad or P.is_managed(p):
        return P.cached(p, "ATTN_PERM32", fn)
    return fn()


# ---- fused attention branch (narrow stages, bf16): LayerNorm -> qkv -> 7x7 window attention -> proj -> residual in one kernel per
# resolution group (esvit_attn_branch_fwd).  Inference passes (the EMA teacher) write nothing but the branch output; a training
# pass also gets the tensors its (unfused) backward reads as side outputs of the same kernel.
ATTN_FUSED = os.environ.get("ESVIT_ATTN_FUSED", "1") != "0"  # (A-B runs switch back to the four-kernel sequence)


def _attn_fused(dt, C, nH, geoms, save, sizes=None):
    """save: a training pass (side outputs on).  Measured on the MI355X (tools/bench_attn_branch.py, B = 128 row counts): without
    side outputs the kernel beats the four-kernel sequence at both widths (C = 96: 226 vs 610 us on the 224-crop rows; C = 192:
    226 vs 299); WITH the 10 B per token-channel of side outputs it still wins at C = 96 (387 + 309 vs ~900 us per block over both
    groups) and loses at C = 192 (292 + 283 vs ~470), so a training pass takes it at C = 96 only"""
    rows, wins = (max(r for r, _ in sizes), max(w for _, w in sizes)) if sizes else (0, 0)  # (token rows, windows) of the largest call
    if not (ATTN_FUSED and all(g.ws == 7 for g in geoms) and ops_module().attn_branch_supported(dt, C, nH, max(g.N for g in geoms), rows, wins)):
        return False
    return C == 96 or not save


# ---- fused MLP branch (narrow stages, bf16): nothing hidden-sized is kept between forward and backward ------------------------
# forward: ONE kernel x1 -> x2 (esvit_mlp_fused_fwd).  backward: esvit_mlp_fused_bwd recomputes LayerNorm + pre-activation,
# produces dL/dx1 (+ its activation-dtype copy) and the operands of the two weight-gradient GEMMs; the LayerNorm parameter
# gradients fall out of the fc1 weight gradient (esvit_ln_fold_finish).
MLP_FUSED_TRAIN = os.environ.get("ESVIT_MLP_FUSED_TRAIN", "1") != "0"  # (A-B runs switch the training path back to the unfused sequence)


# the C = 384 branch: "0" the LayerNorm + two-GEMM sequence everywhere, "1" the fused kernel in inference passes (the EMA teacher), "2" also
# in the training pass (esvit_mlp_fused_fwd_train)
_WIDE = os.environ.get("ESVIT_MLP_WIDE_FUSED", "1")
MLP_WIDE_FUSED = _WIDE != "0"
MLP_WIDE_TRAIN = _WIDE == "2"


def _mlp_wide_train(W1, C, rows):
    """the wide stage's training pass: forward fused (esvit_mlp_fused_fwd_train writes the operands of the unfused backward)"""
    return MLP_WIDE_TRAIN and ops_module().mlp_fused_train_supported(W1.dtype, C, rows)


def _mlp_fused_infer(W1, C):
    return ops_module().mlp_fused_supported(W1.dtype, C) and (MLP_WIDE_FUSED or C != 384)


def _mlp_fused_train(W1, C):
    return MLP_FUSED_TRAIN and ops_module().mlp_fused_supported(W1.dtype, C, backward=True)


def _mlp_branch_bwd(o, x1, gy, dyb, g2, b2, W1, bfc1, params, dp_mlp, dp_out):
    """-> (gx1 fp32, gx1 act copy (scaled by dp_out), dg2, db2, dW1, dbfc1, dW2, dbfc2); dyb: cast(dp_mlp * gy) [M, C] act"""
    g2_p, b2_p, W1_p, bfc1_p, W2_p, bfc2_p = params
    gx1, dyw, xhat, a1g, da1 = o.mlp_fused_bwd(x1, gy, g2, b2, LN_EPS, _mlp_w(W1_p, "MLP_W1_BWD"), _mlp_w(W2_p, "MLP_W2T_BWD"), _mlp_w(W1_p, "MLP_W1T_BWD"), bfc1,
                                               rowscale_mlp=dp_mlp, rowscale_out=dp_out)
    ln2 = _ln_sinks(g2_p, b2_p)
    wsink, bsink = P.grad_out(W1_p), P.grad_out(bfc1_p)

    def wgrads():
        dW2, dbfc2 = _wgrad(dyb, a1g, W2_p, want_bias=True, bias_param=bfc2_p)
        G, dbfc1 = o.linear_wgrad(da1, xhat, out=wsink, want_bias=True, db_out=bsink)  # G = dA^T xhat: LayerNorm folded out
        dW1, dg2, db2 = o.ln_fold_finish(G, dbfc1, W1_p.detach(), g2, b2, gb_out=ln2)
        return dW2, dbfc2, dbfc1, dW1, dg2, db2

    dW2, dbfc2, dbfc1, dW1, dg2, db2 = _side_run(wgrads, dyb, a1g, da1, xhat)
    del a1g, da1, xhat
    return gx1, dyw, _alias(dg2, ln2), _alias(db2, ln2), _alias(dW1, wsink), _alias(dbfc1, bsink), dW2, dbfc2


# ------------------------------------------------------------------------------------------------
# Swin block
# ------------------------------------------------------------------------------------------------
def _block_forward(x, geom, nH, index, dp, prm, wts, save, w1p=None, wattn=None):
    """x fp32 [nB, L, C].  prm: fp32 parameters; wts: activation-dtype weight copies.
    dp: None or (scale_attn [nB], scale_mlp [nB]) DropPath factors."""
    o = ops_module()
    (g1, b1, table, bqkv, bproj, g2, b2, bfc1, bfc2) = prm
    (Wqkv, Wproj, W1, W2) = wts
    nB, L, C = x.shape
    x2d = x.view(nB * L, C)
    scale = (C // nH) ** -0.5
    dp1, dp2 = (None, None) if dp is None else dp
    # everything stays in token order: the attention kernel applies pad/roll/partition through geom.win2tok and
    # injects the qkv bias at zero-pad slots, so no GEMM ever runs on pad rows
    frag = o.new_bias_frag(nH, geom.N, x.device) if save else None  # kept for the backward (no second fill)
    if wattn is not None and _attn_fused(Wqkv.dtype, C, nH, (geom,), save, [(nB * L, nB * geom.nW)]):
        rs1 = None if dp1 is None else dp1.repeat_interleave(L)
        res = o.attn_branch_fwd(x2d, g1, b1, LN_EPS, _perm_w(wattn[0]), bqkv, _perm_w(wattn[1]), bproj, geom.win2tok, L, table, geom.ws, geom.region_ids,
                                geom.nW, geom.N, nH, scale, rowscale=rs1, bias_frag=frag, save=bool(save))
        lse = None
        if save:
            x1, (xw, mean1, rstd1, qkv, ao) = res
        else:
            x1 = res
            xw = mean1 = rstd1 = qkv = ao = None
    else:
        xw, _, mean1, rstd1 = o.layernorm_fwd(x2d, g1, b1, LN_EPS)
        qkv = o.linear_fwd(xw, Wqkv, bqkv)
        ao, lse = o.window_attn_fwd(qkv, bqkv, geom.win2tok, L, table, geom.ws, geom.region_ids, geom.nW, geom.N, nH, scale, bias_frag=frag)
        x1 = o.linear_fwd(ao, Wproj, bproj, residual=x2d, rowscale=dp1, rows_per_sample=L, out_f32=True)
    if (not save and _mlp_fused_infer(W1, C)) or (save and _mlp_fused_train(W1, C)):
        rs2 = None if dp2 is None else dp2.repeat_interleave(L)  # (the fused kernels take per-row DropPath factors)
        x2 = o.mlp_fused_fwd(x1, g2, b2, LN_EPS, W1 if C == 384 else _mlp_w(w1p, "MLP_W1_FWD"), bfc1, W2, bfc2, rowscale=rs2)
        saved = (mean1, rstd1, xw, qkv, ao, x1, lse, frag) if save else None
        return x2.view(nB, L, C), saved
    elif save and _mlp_wide_train(W1, C, nB * L):
        rs2 = None if dp2 is None else dp2.repeat_interleave(L)
        x2, a1, a1g, h, mean2, rstd2 = o.mlp_fused_fwd_train(x1, g2, b2, LN_EPS, W1, bfc1, W2, bfc2, rowscale=rs2)
    else:
        h, _, mean2, rstd2 = o.layernorm_fwd(x1, g2, b2, LN_EPS)
        if save:
            a1g, a1 = o.linear_fwd(h, W1, bfc1, gelu=True, want_preact=True)
        else:
            a1g, a1 = o.linear_fwd(h, W1, bfc1, gelu=True), None
        x2 = o.linear_fwd(a1g, W2, bfc2, residual=x1, rowscale=dp2, rows_per_sample=L, out_f32=True)
    saved = (mean1, rstd1, xw, qkv, ao, x1, mean2, rstd2, h, a1, a1g, lse, frag) if save else None
    return x2.view(nB, L, C), saved


class SwinBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, geom, nH, index, dp, g1, b1, table, Wqkv_p, bqkv, Wproj_p, bproj, g2, b2, W1_p, bfc1, W2_p, bfc2):
        wts = (_weight(Wqkv_p), _weight(Wproj_p), _weight(W1_p), _weight(W2_p))
        x = x.contiguous()
        y, saved = _block_forward(x, geom, nH, index, dp, (g1, b1, table, bqkv, bproj, g2, b2, bfc1, bfc2), wts, True, W1_p, (Wqkv_p, Wproj_p))
        ctx.geom, ctx.nH, ctx.dp = geom, nH, dp
        ctx.wparams = (Wqkv_p, Wproj_p, W1_p, W2_p)
        ctx.mlp_params = (g2, b2, W1_p, bfc1, W2_p, bfc2)
        ctx.fused_mlp = len(saved) == 8
        ctx.save_for_backward(x, index, g1, table, g2, bqkv, b2, bfc1, *wts, *saved)
        return y

    @staticmethod
    def backward(ctx, gy):
        o = ops_module()
        geom, nH, dp = ctx.geom, ctx.nH, ctx.dp
        if ctx.fused_mlp:
            (x, index, g1, table, g2, bqkv, b2, bfc1, Wqkv, Wproj, W1, W2, mean1, rstd1, xw, qkv, ao, x1, lse, frag) = ctx.saved_tensors
        else:
            (x, index, g1, table, g2, bqkv, b2, bfc1, Wqkv, Wproj, W1, W2, mean1, rstd1, xw, qkv, ao, x1, mean2, rstd2, h, a1,
             a1g, lse, frag) = ctx.saved_tensors
        nB, L, C = x.shape
        M = nB * L
        scale = (C // nH) ** -0.5
        dp1, dp2 = (None, None) if dp is None else dp
        gy = gy.contiguous().view(M, C)
        Wqkv_p, Wproj_p, W1_p, W2_p = ctx.wparams
        # ---- MLP branch ----
        dyb = o.gather_cast(gy, M, rowscale=dp2, rows_per_sample=L)
        if ctx.fused_mlp:
            rs2 = None if dp2 is None else dp2.repeat_interleave(L)
            rs1 = None if dp1 is None else dp1.repeat_interleave(L)
            gx1, dyw, dg2, db2, dW1, dbfc1, dW2, dbfc2 = _mlp_branch_bwd(o, x1, gy, dyb, g2, b2, W1, bfc1, ctx.mlp_params, rs2, rs1)
        else:
            dW2, dbfc2 = _wgrad(dyb, a1g, W2_p, want_bias=True)
            da1 = o.linear_dgrad(dyb, W2, gelu_preact=a1)
            dW1, dbfc1 = _wgrad(da1, h, W1_p, want_bias=True)
            dh = o.linear_dgrad(da1, W1)
            # ---- attention branch ---- (the LayerNorm backward also emits the DropPath-scaled activation-dtype copy of gx1)
            gx1, dyw, dg2, db2 = o.layernorm_bwd_cast(dh, x1, mean2, rstd2, g2, g_in=gy, rowscale=dp1, rows_per_sample=L)
        dWproj, dbproj = _wgrad(dyw, ao, Wproj_p, want_bias=True)
        dao = o.linear_dgrad(dyw, Wproj)
        dqkv, dbias_ws, dpad_ws = o.window_attn_bwd(qkv, bqkv, geom.win2tok, L, dao, ao, lse, None, geom.ws, geom.region_ids,
                                                    geom.nW, geom.N, nH, scale, bias_frag=frag)
        dtable = o.relpos_bias_bwd(dbias_ws, index, geom.N, table.shape[0])
        dWqkv, dbqkv = _wgrad(dqkv, xw, Wqkv_p, want_bias=True)
        o.colsum(dpad_ws, out=dbqkv[C:], accumulate=True)  # k/v bias gradient from the zero-pad slots
        dxw = o.linear_dgrad(dqkv, Wqkv)
        gx, dg1, db1 = o.layernorm_bwd(dxw, x.view(M, C), mean1, rstd1, g1, g_in=gx1)
        _side_join()
        return (gx.view(nB, L, C), None, None, None, None, dg1, db1, dtable, dWqkv, dbqkv, dWproj, dbproj, dg2, db2, dW1, dbfc1,
                dW2, dbfc2)


# ---- ragged multi-resolution block: all crops of a step in ONE set of GEMM / LayerNorm launches ---------------------
# LayerNorm, the four GEMMs and the residual adds of a block are row-wise, so the token rows of the 224^2 crops and of the
# 96^2 crops can run through them together; only the window attention depends on the image geometry and is launched once per
# resolution group on its row range.  Compared with one pass per group (swin_transformer.py:729-751) this halves the GEMM /
# LayerNorm launches, removes the gradient-accumulation adds of every parameter used by both passes, halves the split-K
# partial traffic of the weight gradients and gives the small local-crop GEMMs of stages 2-3 full grids.
def _block_forward_multi(X, segs, nH, dp_rows, prm, wts, save, pre=None, next_norm=None, w1p=None, wattn=None):
    """X fp32 [M, C]; segs: tuple of (row0, nB, L, geom); dp_rows: None or (per-row DropPath scale attn [M], mlp [M]).
    pre: (norm1(X) in the activation dtype, mean, rstd) when the previous block's fused MLP kernel already produced them;
    next_norm: (weight, bias) of the NEXT block's norm1 -- the fused MLP kernel then also emits that block's `pre`.
    -> (y, saved, lses, next block's pre or None)"""
    o = ops_module()
    (g1, b1, table, bqkv, bproj, g2, b2, bfc1, bfc2) = prm
    (Wqkv, Wproj, W1, W2) = wts
    M, C = X.shape
    scale = (C // nH) ** -0.5
    dp1, dp2 = (None, None) if dp_rows is None else dp_rows
    fused_attn = wattn is not None and _attn_fused(Wqkv.dtype, C, nH, [sg[3] for sg in segs], save, [(sg[1] * sg[2], sg[1] * sg[3].nW) for sg in segs])
    lses, frags = [], {}
    if fused_attn:
        # the whole branch in one kernel per resolution group; `pre` (the LayerNorm output the previous block's fused MLP kernel may
        # have produced) is not needed: the kernel normalises the rows it loads for the residual anyway
        Wq_p, Wp_p = _perm_w(wattn[0]), _perm_w(wattn[1])
        x1 = torch.empty_like(X)
        if save:
            xw = torch.empty((M, C), dtype=Wqkv.dtype, device=X.device)
            qkv = torch.empty((M, 3 * C), dtype=Wqkv.dtype, device=X.device)
            ao = torch.empty((M, C), dtype=Wqkv.dtype, device=X.device)
            mean1 = torch.empty((M,), dtype=torch.float32, device=X.device)
            rstd1 = torch.empty_like(mean1)
        else:
            xw = qkv = ao = mean1 = rstd1 = None
        for (r0, nB, L, geom) in segs:
            r1 = r0 + nB * L
            first = (geom.ws, geom.N) not in frags
            if first:
                frags[(geom.ws, geom.N)] = o.new_bias_frag(nH, geom.N, X.device)
            sv = (xw[r0:r1], mean1[r0:r1], rstd1[r0:r1], qkv[r0:r1], ao[r0:r1]) if save else False
            o.attn_branch_fwd(X[r0:r1], g1, b1, LN_EPS, Wq_p, bqkv, Wp_p, bproj, geom.win2tok, L, table if first else None, geom.ws, geom.region_ids,
                              geom.nW, geom.N, nH, scale, rowscale=None if dp1 is None else dp1[r0:r1], out=x1[r0:r1],
                              bias_frag=frags[(geom.ws, geom.N)], save=sv)
            lses.append((None, frags[(geom.ws, geom.N)]))
        # (the next block of the stage takes the same route and needs no LayerNorm hand-over)
        next_norm = None
    else:
        if pre is not None:
            xw, mean1, rstd1 = pre
        else:
            xw, _, mean1, rstd1 = o.layernorm_fwd(X, g1, b1, LN_EPS)
        qkv = o.linear_fwd(xw, Wqkv, bqkv)
        ao = torch.empty((M, C), dtype=qkv.dtype, device=X.device)
        for (r0, nB, L, geom) in segs:
            r1 = r0 + nB * L
            # the fragment-order bias is built once per block and step: later groups and the backward reuse it
            first = (geom.ws, geom.N) not in frags
            if first:
                frags[(geom.ws, geom.N)] = o.new_bias_frag(nH, geom.N, X.device)
            _, lse = o.window_attn_fwd(qkv[r0:r1], bqkv, geom.win2tok, L, table if first else None, geom.ws, geom.region_ids, geom.nW, geom.N, nH, scale,
                                       out=ao[r0:r1], bias_frag=frags[(geom.ws, geom.N)])
            lses.append((lse, frags[(geom.ws, geom.N)]))
        x1 = o.linear_fwd(ao, Wproj, bproj, residual=X, rowscale=dp1, rows_per_sample=1, out_f32=True)
    if (not save and _mlp_fused_infer(W1, C)) or (save and _mlp_fused_train(W1, C)):
        # narrow stage: LayerNorm -> fc1 + GELU -> fc2 + residual in one kernel, the hidden activation never reaches HBM (the
        # training pass keeps x1 only: the backward recomputes).  Wide stage (C = 384), inference pass: the same, without the next LayerNorm
        nxt = None
        w1k = W1 if C == 384 else _mlp_w(w1p, "MLP_W1_FWD")
        if next_norm is not None and C != 384:
            x2, nxt = o.mlp_fused_fwd(x1, g2, b2, LN_EPS, w1k, bfc1, W2, bfc2, rowscale=dp2, next_norm=(next_norm[0].detach(), next_norm[1].detach()))
        else:
            x2 = o.mlp_fused_fwd(x1, g2, b2, LN_EPS, w1k, bfc1, W2, bfc2, rowscale=dp2)
        return x2, ((mean1, rstd1, xw, qkv, ao, x1) if save else None), lses, nxt
    elif save and _mlp_wide_train(W1, C, M):
        # wide stage, training pass: one kernel writes x2 and what the unfused backward below reads (LayerNorm output and statistics,
        # pre-activation, GELU) in place of the LayerNorm launch and the two GEMM launches
        x2, a1, a1g, h, mean2, rstd2 = o.mlp_fused_fwd_train(x1, g2, b2, LN_EPS, W1, bfc1, W2, bfc2, rowscale=dp2)
    else:
        h, _, mean2, rstd2 = o.layernorm_fwd(x1, g2, b2, LN_EPS)
        if save:
            a1g, a1 = o.linear_fwd(h, W1, bfc1, gelu=True, want_preact=True)
        else:
            a1g, a1 = o.linear_fwd(h, W1, bfc1, gelu=True), None
        x2 = o.linear_fwd(a1g, W2, bfc2, residual=x1, rowscale=dp2, rows_per_sample=1, out_f32=True)
    saved = (mean1, rstd1, xw, qkv, ao, x1, mean2, rstd2, h, a1, a1g) if save else None
    return x2, saved, lses, None


class SwinBlockMultiFn(torch.autograd.Function):
    """One Swin block over the token rows of several resolution groups.

    Besides y the block returns a SHADOW output ysh (an uninitialised [M, C] tensor of the activation dtype, never read or
    written in the forward).  It exists for the backward: the next block of the stage takes (y, ysh) as inputs and returns,
    as the "gradient" of ysh, cast(prev_scale * dL/dy) -- the activation-dtype, DropPath-scaled copy of dL/dy that this block's
    MLP-branch GEMMs need as their operand -- emitted by the LayerNorm-backward pass that produces dL/dy anyway
    (ops.layernorm_bwd_cast) instead of a separate read-and-cast pass over dL/dy.  If nobody consumes ysh (last block of a
    stage) its gradient arrives as None and the block casts dL/dy itself."""

    @staticmethod
    def forward(ctx, X, Xsh, segs, nH, index, dp_rows, prev_scale, pre, next_norm, g1, b1, table, Wqkv_p, bqkv, Wproj_p, bproj, g2, b2, W1_p, bfc1,
                W2_p, bfc2):
        """pre / next_norm: see _block_forward_multi (plain tuples: no gradient flows through them -- norm1's own backward uses
        the saved statistics).  Third output: the next block's `pre` tuple flattened (xw, mean, rstd), or three None"""
        wts = (_weight(Wqkv_p), _weight(Wproj_p), _weight(W1_p), _weight(W2_p))
        X = X.contiguous()
        y, saved, lses, nxt = _block_forward_multi(X, segs, nH, dp_rows, (g1, b1, table, bqkv, bproj, g2, b2, bfc1, bfc2), wts, True, pre, next_norm, W1_p,
                                                   (Wqkv_p, Wproj_p))
        ctx.segs, ctx.nH, ctx.dp_rows, ctx.lses = segs, nH, dp_rows, lses
        ctx.wparams = (Wqkv_p, Wproj_p, W1_p, W2_p)
        ctx.sparams = (g1, b1, table, bqkv, bproj, g2, b2, bfc1, bfc2)  # small parameters: their gradients go to bucket slots too
        ctx.emit_shadow, ctx.prev_scale = Xsh is not None, prev_scale
        ctx.set_materialize_grads(False)
        ctx.fused_mlp = len(saved) == 6
        ctx.save_for_backward(X, index, g1, table, g2, bqkv, b2, bfc1, *wts, *saved)
        ysh = torch.empty(X.shape, dtype=wts[0].dtype, device=X.device)
        if nxt is None:
            return y, ysh, None, None, None
        ctx.mark_non_differentiable(*nxt)
        return (y, ysh) + tuple(nxt)

    @staticmethod
    def backward(ctx, gy, gysh, *_unused):
        o = ops_module()
        segs, nH, dp_rows = ctx.segs, ctx.nH, ctx.dp_rows
        if ctx.fused_mlp:
            (X, index, g1, table, g2, bqkv, b2, bfc1, Wqkv, Wproj, W1, W2, mean1, rstd1, xw, qkv, ao, x1) = ctx.saved_tensors
        else:
            (X, index, g1, table, g2, bqkv, b2, bfc1, Wqkv, Wproj, W1, W2, mean1, rstd1, xw, qkv, ao, x1, mean2, rstd2, h, a1, a1g) = ctx.saved_tensors
        M, C = X.shape
        scale = (C // nH) ** -0.5
        dp1, dp2 = (None, None) if dp_rows is None else dp_rows
        gy = gy.contiguous()
        Wqkv_p, Wproj_p, W1_p, W2_p = ctx.wparams
        g1_p, b1_p, table_p, bqkv_p, bproj_p, g2_p, b2_p, bfc1_p, bfc2_p = ctx.sparams
        # ---- MLP branch ----
        dyb = gysh.contiguous() if gysh is not None else o.gather_cast(gy, M, rowscale=dp2, rows_per_sample=1)
        if ctx.fused_mlp:
            gx1, dyw, dg2, db2, dW1, dbfc1, dW2, dbfc2 = _mlp_branch_bwd(o, x1, gy, dyb, g2, b2, W1, bfc1,
                                                                          (g2_p, b2_p, W1_p, bfc1_p, W2_p, bfc2_p), dp2, dp1)
        else:
            dW2, dbfc2 = _side_run(lambda: _wgrad(dyb, a1g, W2_p, want_bias=True, bias_param=bfc2_p), dyb, a1g)
            da1 = o.linear_dgrad(dyb, W2, gelu_preact=a1)
            dW1, dbfc1 = _side_run(lambda: _wgrad(da1, h, W1_p, want_bias=True, bias_param=bfc1_p), da1, h)
            dh = o.linear_dgrad(da1, W1)
            ln2 = _ln_sinks(g2_p, b2_p)
            gx1, dyw, dg2, db2 = o.layernorm_bwd_cast(dh, x1, mean2, rstd2, g2, g_in=gy, rowscale=dp1, rows_per_sample=1, gb_out=ln2)
            dg2, db2 = _alias(dg2, ln2), _alias(db2, ln2)
        # ---- attention branch ----
        dWproj, dbproj = _side_run(lambda: _wgrad(dyw, ao, Wproj_p, want_bias=True, bias_param=bproj_p), dyw, ao)
        dao = o.linear_dgrad(dyw, Wproj)
        dqkv = torch.empty_like(qkv)
        tsink = P.grad_out(table_p)
        # the bias-gradient slabs of all groups in one buffer: ONE fold + scatter launch per block (every launch between the large kernels of
        # this chain is ~10 us of step time, profiles/r06_finish_offchain_ab.txt)
        N = segs[0][3].N
        assert all(geom.N == N for (_, _, _, geom) in segs)
        dbuf, dslabs = o.attn_dbias_slabs(N, [nB * geom.nW for (_, nB, _, geom) in segs], nH, qkv.device)
        pads = []
        for (r0, nB, L, geom), (lse, frag), dslab in zip(segs, ctx.lses, dslabs):
            r1 = r0 + nB * L
            _, _, dpad_ws = o.window_attn_bwd(qkv[r0:r1], bqkv, geom.win2tok, L, dao[r0:r1], ao[r0:r1], lse, None, geom.ws, geom.region_ids,
                                              geom.nW, geom.N, nH, scale, dqkv_out=dqkv[r0:r1], bias_frag=frag, dbias_out=dslab)
            pads.append(dpad_ws)
        dtable = o.relpos_bias_bwd(dbuf, index, N, table.shape[0], out=tsink, accumulate=False if tsink is not None else None)
        dtable = _alias(dtable, tsink)
        wsink, bsink = P.grad_out(Wqkv_p), P.grad_out(bqkv_p)

        def wqkv():
            dWqkv, dbqkv = o.linear_wgrad(dqkv, xw, out=wsink, want_bias=True, db_out=bsink)
            for dpad_ws in pads:
                o.colsum(dpad_ws, out=dbqkv[C:], accumulate=True)  # k/v bias gradient from the zero-pad slots
            return dWqkv, dbqkv

        dWqkv, dbqkv = _side_run(wqkv, dqkv, xw, pads)
        dWqkv, dbqkv = _alias(dWqkv, wsink), _alias(dbqkv, bsink)
        dxw = o.linear_dgrad(dqkv, Wqkv)
        ln1 = _ln_sinks(g1_p, b1_p)
        if ctx.emit_shadow:  # the previous block's operand rides along with dL/dX
            gx, gxb, dg1, db1 = o.layernorm_bwd_cast(dxw, X, mean1, rstd1, g1, g_in=gx1, rowscale=ctx.prev_scale, rows_per_sample=1, gb_out=ln1)
        else:
            gx, dg1, db1 = o.layernorm_bwd(dxw, X, mean1, rstd1, g1, g_in=gx1, gb_out=ln1)
            gxb = None
        dg1, db1 = _alias(dg1, ln1), _alias(db1, ln1)
        _side_join()
        return (gx, gxb, None, None, None, None, None, None, None, dg1, db1, dtable, dWqkv, dbqkv, dWproj, dbproj, dg2, db2, dW1, dbfc1, dW2, dbfc2)


def swin_block_multi(X, segs, nH, index, dp_rows, prm_list, shadow=None, prev_scale=None, pre=None, next_norm=None):
    """one Swin block over the token rows of several resolution groups; X fp32 [M, C].  -> (y, shadow of y or None, next block's
    `pre` or None); pass the previous block's shadow and its MLP-branch DropPath row scale to let this block's backward emit that
    block's cast gradient (see SwinBlockMultiFn).  pre / next_norm: the LayerNorm hand-over between the blocks of a stage
    (_block_forward_multi)"""
    if torch.is_grad_enabled() and (X.requires_grad or any(p.requires_grad for p in prm_list)):
        y, ysh, xw, mean, rstd = SwinBlockMultiFn.apply(X, shadow, segs, nH, index, dp_rows, prev_scale, pre, next_norm, *prm_list)
        return y, ysh, (None if xw is None else (xw, mean, rstd))
    g1, b1, table, Wqkv, bqkv, Wproj, bproj, g2, b2, W1, bfc1, W2, bfc2 = prm_list
    wts = (_weight(Wqkv), _weight(Wproj), _weight(W1), _weight(W2))
    y, _, _, nxt = _block_forward_multi(X.contiguous(), segs, nH, dp_rows, (g1, b1, table, bqkv, bproj, g2, b2, bfc1, bfc2), wts, False, pre, next_norm, W1,
                                        (Wqkv, Wproj))
    return y, None, nxt


def swin_block(x, geom, nH, index, dp, prm_list):
    """prm_list: [g1, b1, table, Wqkv, bqkv, Wproj, bproj, g2, b2, W1, bfc1, W2, bfc2] (fp32 parameters)."""
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in prm_list)):
        return SwinBlockFn.apply(x, geom, nH, index, dp, *prm_list)
    g1, b1, table, Wqkv, bqkv, Wproj, bproj, g2, b2, W1, bfc1, W2, bfc2 = prm_list
    wts = (_weight(Wqkv), _weight(Wproj), _weight(W1), _weight(W2))
    y, _ = _block_forward(x.contiguous(), geom, nH, index, dp, (g1, b1, table, bqkv, bproj, g2, b2, bfc1, bfc2), wts, False, W1, (Wqkv, Wproj))
    return y


def swin_block_attention(x, geom, nH, index, prm_list):
    """forward of one block that also returns the softmax tensor (swin_transformer.py:146,152); inference only."""
    o = ops_module()
    g1, b1, table, Wqkv, bqkv, Wproj, bproj, g2, b2, W1, bfc1, W2, bfc2 = prm_list
    nB, L, C = x.shape
    x2d = x.contiguous().view(nB * L, C)
    xw, _, _, _ = o.layernorm_fwd(x2d, g1, b1, LN_EPS)
    qkv = o.linear_fwd(xw, _weight(Wqkv), bqkv)
    _, _, attn = o.window_attn_fwd(qkv, bqkv, geom.win2tok, L, table, geom.ws, geom.region_ids, geom.nW, geom.N, nH, (C // nH) ** -0.5,
                                   want_attn=True)
    return attn


# ------------------------------------------------------------------------------------------------
# PatchEmbed / PatchMerging / final norm / pooling
# ------------------------------------------------------------------------------------------------
class PatchEmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, Wp, bp, g, b, patch):
        o = ops_module()
        E = Wp.shape[0]
        Kc = Wp.shape[1] * patch * patch
        nB, _, S, _ = img.shape
        cols = o.patch_im2col(img.contiguous(), patch, Kc)
        W = _weight(Wp, (E, Kc))
        y = o.linear_fwd(cols, W, bp, out_f32=True)
        ctx.wshape = tuple(Wp.shape)
        if g is None:  # PATCH_NORM False (swin_transformer.py:476-477, 492-493): the projection is the embedding
            ctx.save_for_backward(cols)
            return y.view(nB, (S // patch) ** 2, E)
        x, _, mean, rstd = o.layernorm_fwd(y, g, b, LN_EPS, dtype=torch.float32)
        ctx.save_for_backward(cols, y, mean, rstd, g)
        return x.view(nB, (S // patch) ** 2, E)

    @staticmethod
    def backward(ctx, gx):
        o = ops_module()
        if len(ctx.saved_tensors) == 1:
            (cols,) = ctx.saved_tensors
            M = cols.shape[0]
            dW, dbp = o.linear_wgrad(o.gather_cast(gx.contiguous().view(M, -1), M), cols, want_bias=True)
            return None, dW.view(ctx.wshape), dbp, None, None, None
        cols, y, mean, rstd, g = ctx.saved_tensors
        M, E = y.shape
        dyb, dg, db = o.layernorm_bwd_to_act(gx.contiguous().view(M, E), y, mean, rstd, g)  # (as PatchEmbedMultiFn)
        dW, dbp = o.linear_wgrad(dyb, cols, want_bias=True)
        dW = dW.view(ctx.wshape)
        return None, dW, dbp, dg, db, None


class PatchEmbedMultiFn(torch.autograd.Function):
    """PatchEmbed of several image batches (one per resolution) at once: the im2col images of all groups share one row
    buffer, so the projection GEMM, its LayerNorm and their backward run once.  Returns fp32 token rows [sum_g nB_g L_g, E]
    (group-major, then sample, then token -- the layout Fn.swin_block_multi consumes)."""

    @staticmethod
    def forward(ctx, Wp, bp, g, b, patch, *imgs):
        o = ops_module()
        E = Wp.shape[0]
        Kc = Wp.shape[1] * patch * patch
        rows = [im.shape[0] * (im.shape[2] // patch) ** 2 for im in imgs]  # one entry per crop tensor, in crop order
        cols = torch.empty((sum(rows), Kc), dtype=o.act_dtype(), device=imgs[0].device)
        r0 = 0
        for im, n in zip(imgs, rows):  # (the reference concatenates the crops of a resolution first, swin_transformer.py:741)
            o.patch_im2col(im.contiguous(), patch, Kc, out=cols[r0:r0 + n])
            r0 += n
        y = o.linear_fwd(cols, _weight(Wp, (E, Kc)), bp, out_f32=True)
        ctx.wshape, ctx.n_img = tuple(Wp.shape), len(imgs)
        if g is None:  # PATCH_NORM False
            ctx.save_for_backward(cols)
            return y
        x, _, mean, rstd = o.layernorm_fwd(y, g, b, LN_EPS, dtype=torch.float32)
        ctx.save_for_backward(cols, y, mean, rstd, g)
        return x

    @staticmethod
    def backward(ctx, gx):
        o = ops_module()
        if len(ctx.saved_tensors) == 1:
            (cols,) = ctx.saved_tensors
            M = cols.shape[0]
            dW, dbp = o.linear_wgrad(o.gather_cast(gx.contiguous().view(M, -1), M), cols, want_bias=True)
            return (dW.view(ctx.wshape), dbp, None, None, None) + (None,) * ctx.n_img
        cols, y, mean, rstd, g = ctx.saved_tensors
        M, E = y.shape
        # dL/dy of the norm is only ever the operand of the projection's weight gradient: written once, in the activation dtype
        dya, dg, db = o.layernorm_bwd_to_act(gx.contiguous().view(M, E), y, mean, rstd, g)
        dW, dbp = o.linear_wgrad(dya, cols, want_bias=True)
        return (dW.view(ctx.wshape), dbp, dg, db, None) + (None,) * ctx.n_img


class PatchMergeMultiFn(torch.autograd.Function):
    """PatchMerging over the token rows of several resolution groups: the 2x2 gather + LayerNorm(4C) runs per group (it
    depends on the grid) into one shared row buffer; the reduction GEMM, its dgrad and wgrad run once over all rows.
    X fp32 [M, C], groups: tuple of (row0, nB, H, W) -> fp32 [M/4, 2C] with the groups in the same order.

    The node takes part in the shadow protocol of SwinBlockMultiFn on both sides.  Xsh / prev_scale: the shadow output and the
    MLP-branch DropPath row scale of the stage's last block -- the LayerNorm backward that produces dL/dX also emits
    cast(prev_scale * dL/dX), that block's GEMM operand.  Second output: a shadow of the merged rows for the next stage's first block,
    whose LayerNorm backward hands back cast(dL/dY) -- the operand of this node's reduction GEMMs -- instead of a cast pass over dL/dY."""

    @staticmethod
    def forward(ctx, X, Xsh, prev_scale, groups, g, b, Wr):
        o = ops_module()
        X = X.contiguous()
        M, C = X.shape
        y = torch.empty((M // 4, 4 * C), dtype=o.act_dtype(), device=X.device)
        mean = torch.empty((M // 4,), dtype=torch.float32, device=X.device)
        rstd = torch.empty_like(mean)
        for (r0, nB, H, W) in groups:
            q0, q1 = r0 // 4, (r0 + nB * H * W) // 4
            o.merge_ln_fwd(X[r0:r0 + nB * H * W].view(nB, H * W, C), g, b, LN_EPS, H, W, out=(y[q0:q1], mean[q0:q1], rstd[q0:q1]))
        Wc = _weight(Wr)
        out = o.linear_fwd(y, Wc, None, out_f32=True)
        ctx.save_for_backward(X, y, mean, rstd, g, Wc)
        ctx.groups, ctx.wparam = groups, Wr
        ctx.emit_shadow, ctx.prev_scale = Xsh is not None, prev_scale
        ctx.set_materialize_grads(False)
        outsh = torch.empty(out.shape, dtype=y.dtype, device=X.device) if any(ctx.needs_input_grad) else None
        return out, outsh

    @staticmethod
    def backward(ctx, go, gosh):
        o = ops_module()
        X, y, mean, rstd, g, Wc = ctx.saved_tensors
        M, C = X.shape
        gob = gosh.contiguous() if gosh is not None else o.gather_cast(go.contiguous(), M // 4)
        dWr = _side_run(lambda: _wgrad(gob, y, ctx.wparam), gob, y)
        dy = o.linear_dgrad(gob, Wc)
        dX = torch.empty_like(X)
        dXsh = torch.empty((M, C), dtype=y.dtype, device=X.device) if ctx.emit_shadow else None
        ps = ctx.prev_scale
        gb = torch.empty((2, 4 * C), dtype=torch.float32, device=X.device)  # dgamma | dbeta: the first group writes, later ones accumulate
        for gi, (r0, nB, H, W) in enumerate(ctx.groups):
            r1 = r0 + nB * H * W
            q0, q1 = r0 // 4, r1 // 4
            o.merge_ln_bwd(dy[q0:q1], X[r0:r1].view(nB, H * W, C), mean[q0:q1], rstd[q0:q1], g, H, W, dx_out=dX[r0:r1], gb_out=gb,
                           accumulate=gi > 0, act_out=None if dXsh is None else dXsh[r0:r1], rowscale=None if (ps is None or dXsh is None) else ps[r0:r1])
        _side_join()
        return dX, dXsh, None, None, gb[0], gb[1], dWr


class PatchMergeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, H, W, g, b, Wr):
        o = ops_module()
        x = x.contiguous()
        nB, L, C = x.shape
        y, mean, rstd = o.merge_ln_fwd(x, g, b, LN_EPS, H, W)
        Wc = _weight(Wr)
        out = o.linear_fwd(y, Wc, None, out_f32=True)
        ctx.save_for_backward(x, y, mean, rstd, g, Wc)
        ctx.hw, ctx.wparam = (H, W), Wr
        return out.view(nB, L // 4, 2 * C)

    @staticmethod
    def backward(ctx, go):
        o = ops_module()
        x, y, mean, rstd, g, Wc = ctx.saved_tensors
        H, W = ctx.hw
        rows = y.shape[0]
        gb = o.gather_cast(go.contiguous().view(rows, -1), rows)
        dWr = _wgrad(gb, y, ctx.wparam)
        dy = o.linear_dgrad(gb, Wc)
        dx, dg, db = o.merge_ln_bwd(dy, x, mean, rstd, g, H, W)
        return dx, None, None, dg, db, dWr


class PadTokensFn(torch.autograd.Function):
    """zero-pad a token grid at the bottom / right ([nB, H*W, C] -> [nB, Hp*Wp, C]): F.pad of PatchMerging for odd feature
    maps (swin_transformer.py:406-408); the backward is the crop"""

    @staticmethod
    def forward(ctx, x, H, W, Hp, Wp):
        nB, L, C = x.shape
        ctx.geo = (nB, H, W, Hp, Wp)
        return ops_module().pad_crop_tokens(x.contiguous().view(nB * L, C), nB, H, W, Hp, Wp).view(nB, Hp * Wp, C)

    @staticmethod
    def backward(ctx, gy):
        nB, H, W, Hp, Wp = ctx.geo
        C = gy.shape[-1]
        return ops_module().pad_crop_tokens(gy.contiguous().view(nB * Hp * Wp, C), nB, Hp, Wp, H, W).view(nB, H * W, C), None, None, None, None


class FinalNormFn(torch.autograd.Function):
    """x fp32 [nB, T, C] -> LayerNorm(x) fp32 (the region features the loss matches on stay fp32)."""

    @staticmethod
    def forward(ctx, x, g, b, eps=LN_EPS):
        o = ops_module()
        x = x.contiguous()
        y, _, mean, rstd = o.layernorm_fwd(x.view(-1, x.shape[-1]), g, b, eps, dtype=torch.float32)
        ctx.save_for_backward(x, mean, rstd, g)
        ctx.nparams = (g, b)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, gy):
        o = ops_module()
        x, mean, rstd, g = ctx.saved_tensors
        C = x.shape[-1]
        sinks = _ln_sinks(*ctx.nparams)
        dx, dg, db = o.layernorm_bwd(gy.contiguous().view(-1, C), x.view(-1, C), mean, rstd, g, gb_out=sinks)
        return dx.view(x.shape), _alias(dg, sinks), _alias(db, sinks), None


class TokenMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        o = ops_module()
        m, _ = o.token_mean_fwd(x.contiguous())
        ctx.T = x.shape[1]
        return m

    @staticmethod
    def backward(ctx, gm):
        return ops_module().token_mean_bwd(gm.contiguous(), None, ctx.T)


# ------------------------------------------------------------------------------------------------
# DINOHead
# ------------------------------------------------------------------------------------------------
def _last_layer_weight(v, g):
    o = ops_module()
    if v.requires_grad or g.requires_grad:
        return P.cached(v, "wn", lambda: o.weightnorm_fwd(v.detach(), g.detach()))
    return o.weightnorm_fwd(v.detach(), g.detach())


def _last_logits(o, z, w, stats):
    """logits of the weight-normed last layer (vision_transformer.py:418); with stats = (inv_temp, center) and a shape the GEMM's
    statistics epilogue covers also their softmax row statistics -> (logits, row_max | None, row_lse | None)"""
    if stats is not None and o.row_stats_supported(z.dtype, z.shape[0], w.shape[0]):
        return o.linear_fwd(z, w, row_stats=stats)
    return o.linear_fwd(z, w), None, None


def _head_forward(x, prm, save, stats=None):
    o = ops_module()
    W1p, b1, W2p, b2, W3p, b3, v, g = prm
    W1, W2, W3 = _weight(W1p), _weight(W2p), _weight(W3p)
    xa = o.cast_to_act(x.contiguous())
    if save:
        h1g, h1 = o.linear_fwd(xa, W1, b1, gelu=True, want_preact=True)
        h2g, h2 = o.linear_fwd(h1g, W2, b2, gelu=True, want_preact=True)
    else:
        h1g, h1 = o.linear_fwd(xa, W1, b1, gelu=True), None
        h2g, h2 = o.linear_fwd(h1g, W2, b2, gelu=True), None
    h3 = o.linear_fwd(h2g, W3, b3)
    z, inv = o.l2norm_fwd(h3)
    w, winv = _last_layer_weight(v, g)
    logits = _last_logits(o, z, w, stats)
    return logits, (W1, W2, W3, xa, h1, h1g, h2, h2g, z, inv, w, winv)


class DinoHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stats, W1p, b1, W2p, b2, W3p, b3, v, g):
        (logits, mx, lse), saved = _head_forward(x, (W1p, b1, W2p, b2, W3p, b3, v, g), True, stats)
        ctx.save_for_backward(v, g, *saved)
        ctx.need_dg = g.requires_grad
        ctx.wparams = (W1p, W2p, W3p, v)
        ctx.bparams = (b1, b2, b3)
        if mx is not None:
            ctx.mark_non_differentiable(mx, lse)
        return logits, mx, lse

    @staticmethod
    def backward(ctx, dlogits, _gmx=None, _glse=None):
        o = ops_module()
        v, g, W1, W2, W3, xa, h1, h1g, h2, h2g, z, inv, w, winv = ctx.saved_tensors
        dlogits = dlogits.contiguous()
        W1p, W2p, W3p, vp = ctx.wparams
        b1p, b2p, b3p = ctx.bparams
        dz = o.linear_dgrad(dlogits, w)
        sink = P.grad_out(vp)

        def wlast():
            dw = o.linear_wgrad(dlogits, z)
            return o.weightnorm_bwd(dw, v, g, winv, ctx.need_dg, dv_out=sink)

        dv, dg = _side_run(wlast, dlogits, z)
        if sink is not None:
            dv = dv.detach()  # a fresh alias of the bucket slot (see _wgrad)
        dh3 = o.l2norm_bwd(dz, z, inv)
        dW3, db3 = _side_run(lambda: _wgrad(dh3, h2g, W3p, want_bias=True, bias_param=b3p), dh3, h2g)
        dh2 = o.linear_dgrad(dh3, W3, gelu_preact=h2)
        dW2, db2 = _side_run(lambda: _wgrad(dh2, h1g, W2p, want_bias=True, bias_param=b2p), dh2, h1g)
        dh1 = o.linear_dgrad(dh2, W2, gelu_preact=h1)
        dW1, db1 = _side_run(lambda: _wgrad(dh1, xa, W1p, want_bias=True, bias_param=b1p), dh1, xa)
        dx = o.linear_dgrad(dh1, W1, out_f32=True)
        _side_join()
        return dx, None, dW1, db1, dW2, db2, dW3, db3, dv, dg


def dino_head(x, prm, stats=None):
    """-> (logits, row_max, row_lse): the statistics are None unless stats = (inv_temp, center) was given and the shape allows"""
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in prm)):
        return DinoHeadFn.apply(x, stats, *prm)
    return _head_forward(x, prm, False, stats)[0]


class DinoHeadNFn(torch.autograd.Function):
    """DINOHead with any number of Linear layers (vision_transformer.py:388-402, nlayers != 3, no BatchNorm): nlayers = 1 is one
    Linear into the bottleneck, otherwise Linear + GELU (nlayers - 1 times) and a last Linear; then l2-normalise and the weight-normed
    last layer as in DinoHeadFn.  params = (W_1, b_1, ..., W_L, b_L, weight_v, weight_g)."""

    @staticmethod
    def forward(ctx, x, stats, *params):
        o = ops_module()
        L = (len(params) - 2) // 2
        Wp, bp = params[0:2 * L:2], params[1:2 * L:2]
        v, g = params[2 * L], params[2 * L + 1]
        Ws = [_weight(w) for w in Wp]
        h = o.cast_to_act(x.contiguous())
        acts, pres = [h], []
        for i in range(L - 1):
            hg, hpre = o.linear_fwd(acts[-1], Ws[i], bp[i], gelu=True, want_preact=True)
            acts.append(hg)
            pres.append(hpre)
        hL = o.linear_fwd(acts[-1], Ws[L - 1], bp[L - 1])
        z, inv = o.l2norm_fwd(hL)
        w, winv = _last_layer_weight(v, g)
        logits, mx, lse = _last_logits(o, z, w, stats)
        ctx.L = L
        ctx.need_dg = g.requires_grad
        ctx.wparams, ctx.bparams, ctx.vparam = tuple(Wp), tuple(bp), v
        ctx.save_for_backward(v, g, z, inv, w, winv, *Ws, *acts, *pres)
        if mx is not None:
            ctx.mark_non_differentiable(mx, lse)
        return logits, mx, lse

    @staticmethod
    def backward(ctx, dlogits, _gmx=None, _glse=None):
        o = ops_module()
        L = ctx.L
        t = ctx.saved_tensors
        v, g, z, inv, w, winv = t[:6]
        Ws, acts, pres = t[6:6 + L], t[6 + L:6 + 2 * L], t[6 + 2 * L:]
        dlogits = dlogits.contiguous()
        dz = o.linear_dgrad(dlogits, w)
        dw = o.linear_wgrad(dlogits, z)
        sink = P.grad_out(ctx.vparam)
        dv, dg = o.weightnorm_bwd(dw, v, g, winv, ctx.need_dg, dv_out=sink)
        if sink is not None:
            dv = dv.detach()
        dh = o.l2norm_bwd(dz, z, inv)
        grads = [None] * (2 * L)
        for i in range(L - 1, -1, -1):
            grads[2 * i], grads[2 * i + 1] = _wgrad(dh, acts[i], ctx.wparams[i], want_bias=True, bias_param=ctx.bparams[i])
            if i > 0:
                dh = o.linear_dgrad(dh, Ws[i], gelu_preact=pres[i - 1])
        dx = o.linear_dgrad(dh, Ws[0], out_f32=True)
        return (dx, None) + tuple(grads) + (dv, dg)


def dino_head_n(x, lin_params, v, g, stats=None):
    """lin_params: [(W, b), ...] of the head's Linear layers in order -> (logits, row_max | None, row_lse | None)"""
    flat = [p for wb in lin_params for p in wb]
    return DinoHeadNFn.apply(x, stats, *flat, v, g)


class ApeAddFn(torch.autograd.Function):
    """x + absolute_pos_embed (swin_transformer.py:680-681, USE_APE): x fp32 [nB, L, C], ape fp32 [1, L, C].  Viewed as
    [1, nB, L*C] the broadcast over the images is the token-broadcast of esvit_token_mean_bwd (x_b + nB*ape / nB), and the
    parameter gradient, the sum over the images, is nB times esvit_token_mean_fwd -- no new kernel."""

    @staticmethod
    def forward(ctx, x, ape):
        o = ops_module()
        nB, L, C = x.shape
        if tuple(ape.shape) != (1, L, C):
            raise ValueError("absolute_pos_embed %s does not fit %d tokens: USE_APE supports crops of the construction resolution only "
                             "(the reference's broadcast add fails the same way)" % (tuple(ape.shape), L))
        n = torch.full((), float(nB), dtype=torch.float32, device=x.device)
        g = o.scale_inplace(ape.detach().clone().view(1, L * C), n)
        ctx.n = n
        return o.token_mean_bwd(g, x.contiguous().view(1, nB, L * C), nB).view(nB, L, C)

    @staticmethod
    def backward(ctx, gy):
        o = ops_module()
        nB, L, C = gy.shape
        dape, _ = o.token_mean_fwd(gy.contiguous().view(1, nB, L * C), dtype=torch.float32)
        return gy, o.scale_inplace(dape, ctx.n).view(1, L, C)


# ---- DINOHead(use_bn=True): Linear -> BatchNorm1d -> GELU (x2) -> Linear -> l2-normalise -> weight-normed last layer
# (vision_transformer.py:391-402, 414-418).  The Linear writes its pre-norm output d; one pass gives the batch sums, the
# [C]-sized coefficient kernel turns them into y = a d + shift and updates the running statistics, and GELU rides on the
# affine pass.  Statistics are summed over the ranks (main_esvit.py:365-379 converts every BatchNorm to SyncBatchNorm).
HEAD_BN_EPS = 1e-5
HEAD_BN_MOMENTUM = 0.1


def _head_bn_coef(o, d, st, gam, bet):
    """-> (coef [4, C], n): batch statistics in training mode, the running ones in eval mode"""
    if st.get("eval"):
        return o.bn_eval_coeffs(st["eval_mean"], st["eval_var"], gam, bet, HEAD_BN_EPS), float(d.shape[0])
    sums = o.col_sums2(d, d)
    n = float(d.shape[0] * _allreduce_stats(sums, st.get("group")))
    coef = o.bn_fwd_coeffs(sums, n, gam, bet, HEAD_BN_EPS, HEAD_BN_MOMENTUM, st.get("running_mean"), st.get("running_var"))
    if st.get("num_batches_tracked") is not None:
        st["num_batches_tracked"].add_(1)
    return coef, n


class DinoHeadBnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, st1, st2, stats, W1p, b1, g1, be1, W2p, b2, g2, be2, W3p, b3, v, g):
        o = ops_module()
        W1, W2, W3 = _weight(W1p), _weight(W2p), _weight(W3p)
        xa = o.cast_to_act(x.contiguous())
        gam1, bet1, gam2, bet2 = (t.detach().contiguous() for t in (g1, be1, g2, be2))
        d1 = o.linear_fwd(xa, W1, b1)
        coef1, n1 = _head_bn_coef(o, d1, st1, gam1, bet1)
        h1 = o.col_affine2(d1, coef1[0], coef1[1], act=1)
        d2 = o.linear_fwd(h1, W2, b2)
        coef2, n2 = _head_bn_coef(o, d2, st2, gam2, bet2)
        h2 = o.col_affine2(d2, coef2[0], coef2[1], act=1)
        h3 = o.linear_fwd(h2, W3, b3)
        z, inv = o.l2norm_fwd(h3)
        w, winv = _last_layer_weight(v, g)
        logits, mx, lse = _last_logits(o, z, w, stats)
        ctx.save_for_backward(v, g, W1, W2, W3, xa, d1, coef1, gam1, h1, d2, coef2, gam2, h2, z, inv, w, winv)
        ctx.meta = (n1, n2, st1.get("group"), bool(st1.get("eval")), g.requires_grad)
        ctx.wparams = (W1p, W2p, W3p, v)
        if mx is not None:
            ctx.mark_non_differentiable(mx, lse)
        return logits, mx, lse

    @staticmethod
    def _bn_gelu_bwd(o, dh, d, coef, gam, n, group, eval_bn):
        """dh = dL/d GELU(BN(d)) -> (dL/dd, dgamma, dbeta)"""
        du = o.col_affine2(d, coef[0], coef[1], x2=dh, act=2)     # through the GELU, at the rebuilt BN output
        red = o.bn_bwd_local(o.col_sums2(du, d), coef)            # (sum du, sum du*xhat) of this rank = (d beta, d gamma)
        dbet, dgam = red[0].clone(), red[1].clone()
        if eval_bn:
            abc = o.bn_bwd_coeffs(None, n, gam, coef)
        else:
            _allreduce_stats(red, group)
            abc = o.bn_bwd_coeffs(red, n, gam, coef)
        return o.col_affine2(du, abc[0], abc[2], d, abc[1]), dgam, dbet

    @staticmethod
    def backward(ctx, dlogits, _gmx=None, _glse=None):
        o = ops_module()
        v, g, W1, W2, W3, xa, d1, coef1, gam1, h1, d2, coef2, gam2, h2, z, inv, w, winv = ctx.saved_tensors
        n1, n2, group, eval_bn, need_dg = ctx.meta
        W1p, W2p, W3p, vp = ctx.wparams
        dlogits = dlogits.contiguous()
        dz = o.linear_dgrad(dlogits, w)
        dw = o.linear_wgrad(dlogits, z)
        sink = P.grad_out(vp)
        dv, dg = o.weightnorm_bwd(dw, v, g, winv, need_dg, dv_out=sink)
        if sink is not None:
            dv = dv.detach()
        dh3 = o.l2norm_bwd(dz, z, inv)
        dW3, db3 = _wgrad(dh3, h2, W3p, want_bias=True)
        dd2, dg2, dbe2 = DinoHeadBnFn._bn_gelu_bwd(o, o.linear_dgrad(dh3, W3), d2, coef2, gam2, n2, group, eval_bn)
        dW2, db2 = _wgrad(dd2, h1, W2p, want_bias=True)
        dd1, dg1, dbe1 = DinoHeadBnFn._bn_gelu_bwd(o, o.linear_dgrad(dd2, W2), d1, coef1, gam1, n1, group, eval_bn)
        dW1, db1 = _wgrad(dd1, xa, W1p, want_bias=True)
        dx = o.linear_dgrad(dd1, W1, out_f32=True)
        return dx, None, None, None, dW1, db1, dg1, dbe1, dW2, db2, dg2, dbe2, dW3, db3, dv, dg


class DinoHeadBnNFn(torch.autograd.Function):
    """DINOHead(use_bn=True) with any number of layers (vision_transformer.py:391-402, nlayers != 3): (Linear -> BatchNorm1d -> GELU)
    nlayers - 1 times, a Linear into the bottleneck, l2-normalise, the weight-normed last layer -- the passes of DinoHeadBnFn in a
    loop.  params = (W_1, b_1, bn_1.weight, bn_1.bias, ..., W_L, b_L, weight_v, weight_g); bn_states: one dict per BatchNorm."""

    @staticmethod
    def forward(ctx, x, bn_states, stats, *params):
        o = ops_module()
        L = (len(params) - 2 + 2) // 4                      # hidden layers carry four tensors, the last Linear two
        hid = [params[4 * i:4 * i + 4] for i in range(L - 1)]
        Wl, bl = params[4 * (L - 1)], params[4 * (L - 1) + 1]
        v, g = params[-2], params[-1]
        h = o.cast_to_act(x.contiguous())
        saved, meta = [], []
        for (Wp, b, gam, bet), st in zip(hid, bn_states):
            W = _weight(Wp)
            gam_, bet_ = gam.detach().contiguous(), bet.detach().contiguous()
            d = o.linear_fwd(h, W, b)
            coef, n = _head_bn_coef(o, d, st, gam_, bet_)
            saved += [W, h, d, coef, gam_]
            meta.append(n)
            h = o.col_affine2(d, coef[0], coef[1], act=1)
        WL = _weight(Wl)
        hL = o.linear_fwd(h, WL, bl)
        z, inv = o.l2norm_fwd(hL)
        w, winv = _last_layer_weight(v, g)
        logits, mx, lse = _last_logits(o, z, w, stats)
        ctx.L, ctx.ns = L, meta
        ctx.group, ctx.eval_bn = (bn_states[0].get("group"), bool(bn_states[0].get("eval"))) if bn_states else (None, False)
        ctx.need_dg = g.requires_grad
        ctx.wparams = tuple(hp[0] for hp in hid) + (Wl,)
        ctx.vparam = v
        ctx.save_for_backward(v, g, z, inv, w, winv, WL, h, *saved)
        if mx is not None:
            ctx.mark_non_differentiable(mx, lse)
        return logits, mx, lse

    @staticmethod
    def backward(ctx, dlogits, _gmx=None, _glse=None):
        o = ops_module()
        L = ctx.L
        t = ctx.saved_tensors
        v, g, z, inv, w, winv, WL, hlast = t[:8]
        per = [t[8 + 5 * i:8 + 5 * i + 5] for i in range(L - 1)]   # (W, input, pre-norm output, coef, gamma) of hidden layer i
        dlogits = dlogits.contiguous()
        dz = o.linear_dgrad(dlogits, w)
        dw = o.linear_wgrad(dlogits, z)
        sink = P.grad_out(ctx.vparam)
        dv, dg = o.weightnorm_bwd(dw, v, g, winv, ctx.need_dg, dv_out=sink)
        if sink is not None:
            dv = dv.detach()
        dh = o.l2norm_bwd(dz, z, inv)
        dWl, dbl = _wgrad(dh, hlast, ctx.wparams[L - 1], want_bias=True)
        dh = o.linear_dgrad(dh, WL)
        grads = [None] * (4 * (L - 1))
        for i in range(L - 2, -1, -1):
            W, hin, d, coef, gam = per[i]
            dd, dgam, dbet = DinoHeadBnFn._bn_gelu_bwd(o, dh, d, coef, gam, ctx.ns[i], ctx.group, ctx.eval_bn)
            dW, db = _wgrad(dd, hin, ctx.wparams[i], want_bias=True)
            grads[4 * i:4 * i + 4] = [dW, db, dgam, dbet]
            dh = o.linear_dgrad(dd, W, out_f32=(i == 0))
        return (dh, None, None) + tuple(grads) + (dWl, dbl, dv, dg)


def dino_head_bn_n(x, bn_states, hidden, last, v, g, stats=None):
    """hidden: [(W, b, bn.weight, bn.bias), ...]; last: (W, b) of the Linear into the bottleneck -> (logits, row_max, row_lse)"""
    flat = [p for h in hidden for p in h]
    return DinoHeadBnNFn.apply(x, list(bn_states), stats, *flat, *last, v, g)


def dino_head_bn(x, st1, st2, prm, stats=None):
    """prm = (W1, b1, bn1.weight, bn1.bias, W2, b2, bn2.weight, bn2.bias, W3, b3, weight_v, weight_g) -> (logits, row_max, row_lse)"""
    return DinoHeadBnFn.apply(x, st1, st2, stats, *prm)


# ------------------------------------------------------------------------------------------------
# CvT (cvt_v4_transformer.py; BASELINE config 5).  Activations stay token-major NHWC, so the reference's
# 'b c h w <-> b h w c' rearranges around every LayerNorm do not exist here.
# ------------------------------------------------------------------------------------------------
CVT_LN_EPS = 1e-5   # get_cls_model: norm_layer=partial(LayerNorm, eps=1e-5)  (cvt_v4_transformer.py:694)
BN_EPS = 1e-5       # nn.BatchNorm2d defaults (cvt_v4_transformer.py:95)
BN_MOMENTUM = 0.1


def _conv_weight_matrix(Wp):
    """conv weight [E, Cin, k, k] -> activation-dtype GEMM operand [E, Kpad], column order (ky, kx, c), zero tail"""
    def make():
        E, Cin, k, _ = Wp.shape
        K = k * k * Cin
        Kpad = -(-K // 8) * 8
        m = torch.zeros((E, Kpad), dtype=torch.float32, device=Wp.device)
        m[:, :K] = Wp.detach().permute(0, 2, 3, 1).reshape(E, K)
        return ops_module().cast_to_act(m)
    if Wp.requires_grad or P.is_managed(Wp):
        return P.cached(Wp, "convk", make)
    return make()


def _pad_tokens(t, nB, H, W, Hp, Wp):
    """[nB*H*W, C] -> [nB*Hp*Wp, C] with zero rows/cols appended bottom/right (F.pad of cvt_v4_transformer.py:173)"""
    if Hp == H and Wp == W:
        return t
    return ops_module().pad_crop_tokens(t, nB, H, W, Hp, Wp)


def _crop_tokens(t, nB, H, W, Hp, Wp):
    if Hp == H and Wp == W:
        return t
    return ops_module().pad_crop_tokens(t, nB, Hp, Wp, H, W)


def _allreduce_stats(t, group):
    """SyncBatchNorm: batch statistics are sums over every rank's positions (main_esvit.py:365-379 converts the BN layers)"""
    import torch.distributed as dist
    if group is not False and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, group=group)
        return dist.get_world_size(group)
    return 1


class ConvEmbedFn(torch.autograd.Function):
    """ConvEmbed (cvt_v4_transformer.py:349-382): conv k x k / stride / pad as im2col + GEMM, then LayerNorm.
    src: fp32 NCHW images (first stage) or fp32 tokens [nB, H*W, Cin]; returns fp32 tokens [nB, Ho*Wo, E]."""

    @staticmethod
    def forward(ctx, src, geo, Wp, bp, g, b):
        o = ops_module()
        nchw, nB, H, W, Cin, k, stride, pad = geo[:8]
        eps = geo[8] if len(geo) > 8 else CVT_LN_EPS  # (Vision Longformer's patch embeddings: eps 1e-6)
        src = src.contiguous()
        if nchw:
            cols = o.conv_im2col(src, True, nB, H, W, Cin, k, stride, pad)
        else:
            cols = o.conv_im2col(o.gather_cast(src.view(nB * H * W, Cin), nB * H * W), False, nB, H, W, Cin, k, stride, pad)
        Wk = _conv_weight_matrix(Wp)
        y = o.linear_fwd(cols, Wk, bp, out_f32=True)
        t, _, mean, rstd = o.layernorm_fwd(y, g, b, eps, dtype=torch.float32)
        ctx.geo = geo
        ctx.wshape = tuple(Wp.shape)
        ctx.save_for_backward(cols, Wk, y, mean, rstd, g)
        Ho, Wo = o.conv_out_size(H, k, stride, pad), o.conv_out_size(W, k, stride, pad)
        return t.view(nB, Ho * Wo, -1)

    @staticmethod
    def backward(ctx, gt):
        o = ops_module()
        cols, Wk, y, mean, rstd, g = ctx.saved_tensors
        nchw, nB, H, W, Cin, k, stride, pad = ctx.geo[:8]
        E = y.shape[1]
        dy, dg, db = o.layernorm_bwd(gt.contiguous().view(-1, E), y, mean, rstd, g)
        dyb = o.gather_cast(dy, dy.shape[0])
        dWk, dbp = o.linear_wgrad(dyb, cols, want_bias=True)
        K = k * k * Cin
        dWp = dWk[:, :K].reshape(E, k, k, Cin).permute(0, 3, 1, 2).contiguous()
        dsrc = None
        if not nchw and ctx.needs_input_grad[0]:
            dcols = o.linear_dgrad(dyb, Wk)
            dsrc = o.conv_col2im(dcols, nB, H, W, Cin, k, stride, pad).view(nB, H * W, Cin)
        return dsrc, None, dWp, dbp, dg, db


class ConvBnReluFn(torch.autograd.Function):
    """one unit of the residual stem (cvt_v4_transformer.py:385-430): Conv2d(3x3, no bias) -> BatchNorm2d -> ReLU as im2col + GEMM,
    per-channel sums, one affine + ReLU pass.  src: fp32 NCHW images or activation-dtype tokens [nB*H*W, Cin]; returns activation-dtype
    tokens [nB*Ho*Wo, E].  BatchNorm as in CvtAttnFn (batch statistics summed over the ranks in train mode, running statistics in eval)."""

    @staticmethod
    def forward(ctx, src, geo, bn_state, Wp, bn_g, bn_b):
        o = ops_module()
        nchw, nB, H, W, Cin, k, stride, pad = geo
        src = src.contiguous()
        cols = o.conv_im2col(src if nchw else src.view(nB * H * W, Cin), nchw, nB, H, W, Cin, k, stride, pad)
        Wk = _conv_weight_matrix(Wp)
        d = o.linear_fwd(cols, Wk, None)
        rows = d.shape[0]
        eval_bn = bool(bn_state.get("eval"))
        gam, bet = bn_g.detach().contiguous(), bn_b.detach().contiguous()
        if eval_bn:
            n = float(rows)
            coef = o.bn_eval_coeffs(bn_state["eval_mean"], bn_state["eval_var"], gam, bet, BN_EPS)
        else:
            sums = o.col_sums2(d, d)
            n = float(rows * _allreduce_stats(sums, bn_state.get("group")))
            coef = o.bn_fwd_coeffs(sums, n, gam, bet, BN_EPS, BN_MOMENTUM, bn_state.get("running_mean"), bn_state.get("running_var"))
            if bn_state.get("num_batches_tracked") is not None:
                bn_state["num_batches_tracked"].add_(1)
        y = o.col_affine2(d, coef[0], coef[1], act=3)
        ctx.geo, ctx.n, ctx.eval_bn, ctx.group, ctx.wshape = geo, n, eval_bn, bn_state.get("group"), tuple(Wp.shape)
        ctx.save_for_backward(cols, Wk, d, coef, gam)
        return y

    @staticmethod
    def backward(ctx, gy):
        o = ops_module()
        cols, Wk, d, coef, gam = ctx.saved_tensors
        nchw, nB, H, W, Cin, k, stride, pad = ctx.geo
        E = d.shape[1]
        gy = gy.contiguous()
        if gy.dtype != d.dtype:
            gy = o.cast_to_act(gy.float())
        dz = o.col_affine2(d, coef[0], coef[1], gy, act=4)        # through the ReLU, its argument rebuilt from the conv output
        red = o.bn_bwd_local(o.col_sums2(dz, d), coef)            # (d beta, d gamma) of this rank
        dbet, dgam = red[0].clone(), red[1].clone()
        if ctx.eval_bn:
            abc = o.bn_bwd_coeffs(None, ctx.n, gam, coef)
        else:
            _allreduce_stats(red, ctx.group)
            abc = o.bn_bwd_coeffs(red, ctx.n, gam, coef)
        dd = o.col_affine2(dz, abc[0], abc[2], d, abc[1])
        dWk = o.linear_wgrad(dd, cols)
        K = k * k * Cin
        dWp = dWk[:, :K].reshape(E, k, k, Cin).permute(0, 3, 1, 2).contiguous()
        dsrc = None
        if not nchw and ctx.needs_input_grad[0]:
            dsrc = o.conv_col2im(o.linear_dgrad(dd, Wk), nB, H, W, Cin, k, stride, pad)
            dsrc = o.gather_cast(dsrc, dsrc.shape[0])
        return dsrc, None, None, dWp, dgam, dbet


class CvtAttnFn(torch.autograd.Function):
    """x + DropPath(Attention(LayerNorm(x)))  (cvt_v4_transformer.py:331-336, 49-58, 108-220): LN -> zero-pad the grid to a
    multiple of the window -> depthwise 3x3 -> BatchNorm (batch statistics, synchronised across ranks) -> 1x1 to q|k|v ->
    windowed attention (w = min(window, H, W), scale = dim ** -0.5) -> crop -> 1x1 proj.  Optional (s1_rpe.yaml / s1_shift.yaml):
    a relative-position bias table (`table_p`, `index`) and the shifted-window mask of a half-window shift on the UNROLLED map
    (`shift`: cvt_v4_transformer.py:291-329 builds the Swin mask, :332 hands it to every block, nothing rolls)."""

    @staticmethod
    def forward(ctx, x, H, W, nH, window, dp, bn_state, g1, b1, dw_w, bn_g, bn_b, pw_Wp, pw_b, proj_Wp, proj_b, table_p=None, index=None,
                shift=False):
        o = ops_module()
        nB, L, C = x.shape
        x = x.contiguous()
        x2d = x.view(nB * L, C)
        w = min(window, H, W)
        if (table_p is not None or shift) and w != window:
            # the reference adds a [window^2, window^2] bias / mask to [w^2, w^2] scores here and fails on the shapes
            raise RuntimeError("CvT relative-position bias / shift mask on a %dx%d map: smaller than the %dx%d window" % (H, W, window, window))
        if shift and (H % w or W % w):
            # the reference's masked branch overwrites its map height with the head count (cvt_v4_transformer.py:201) and then crops
            # the padded output with it: only maps that need no padding come through
            raise RuntimeError("CvT shift mask on a %dx%d map: not a multiple of the %dx%d window" % (H, W, window, window))
        Hp, Wp = -(-H // w) * w, -(-W // w) * w
        xn, _, mean1, rstd1 = o.layernorm_fwd(x2d, g1, b1, CVT_LN_EPS)
        xp = _pad_tokens(xn, nB, H, W, Hp, Wp)
        dw9 = dw_w.detach().reshape(C, 9).contiguous()
        d = o.dwconv3x3(xp, dw9, nB, Hp, Wp)
        rows = nB * Hp * Wp
        eval_bn = bool(bn_state.get("eval"))
        gam, bet = bn_g.detach().contiguous(), bn_b.detach().contiguous()
        if eval_bn:  # inference: the running statistics (nn.BatchNorm2d in eval mode)
            n = float(rows)
            coef = o.bn_eval_coeffs(bn_state["eval_mean"], bn_state["eval_var"], gam, bet, BN_EPS)
        else:        # BatchNorm2d, training statistics over every position of the (padded) map on every rank
            sums = o.col_sums2(d, d)
            n = float(rows * _allreduce_stats(sums, bn_state.get("group")))  # every rank runs the same per-GPU batch
            coef = o.bn_fwd_coeffs(sums, n, gam, bet, BN_EPS, BN_MOMENTUM, bn_state.get("running_mean"), bn_state.get("running_var"))
            if bn_state.get("num_batches_tracked") is not None:
                bn_state["num_batches_tracked"].add_(1)
        bnout = o.col_affine2(d, coef[0], coef[1])
        Wpw, Wproj = _weight(pw_Wp, (3 * C, C)), _weight(proj_Wp, (C, C))
        qkv = o.linear_fwd(bnout, Wpw, pw_b)
        geom = geometry(Hp, Wp, w, 0, x.device)
        table = _zero_table(w, nH, x.device) if table_p is None else table_p.detach()
        regions = geometry(Hp, Wp, w, w // 2, x.device).region_ids if shift else None
        scale = float(C) ** -0.5
        ao, lse = o.window_attn_fwd(qkv, pw_b, geom.win2tok, Hp * Wp, table, w, regions, geom.nW, geom.N, nH, scale)
        aoc = _crop_tokens(ao, nB, H, W, Hp, Wp)
        x1 = o.linear_fwd(aoc, Wproj, proj_b, residual=x2d, rowscale=dp, rows_per_sample=L, out_f32=True)
        ctx.meta = (H, W, Hp, Wp, w, nH, scale, dp, bn_state.get("group"), eval_bn, table_p is not None, shift)
        ctx.n = n
        ctx.save_for_backward(x, mean1, rstd1, g1, xp, dw9, d, coef, gam, bnout, Wpw, pw_b, qkv, ao, aoc, Wproj, lse, table, index)
        return x1.view(nB, L, C)

    @staticmethod
    def backward(ctx, gy):
        o = ops_module()
        x, mean1, rstd1, g1, xp, dw9, d, coef, gam, bnout, Wpw, pw_b, qkv, ao, aoc, Wproj, lse, table, index = ctx.saved_tensors
        n = ctx.n
        H, W, Hp, Wp, w, nH, scale, dp, group, eval_bn, has_table, shift = ctx.meta
        nB, L, C = x.shape
        M = nB * L
        gy = gy.contiguous().view(M, C)
        dyb = o.gather_cast(gy, M, rowscale=dp, rows_per_sample=L)
        dWproj, dbproj = _side_run(lambda: o.linear_wgrad(dyb, aoc, want_bias=True), dyb, aoc)
        dao = _pad_tokens(o.linear_dgrad(dyb, Wproj), nB, H, W, Hp, Wp)
        geom = geometry(Hp, Wp, w, 0, x.device)
        regions = geometry(Hp, Wp, w, w // 2, x.device).region_ids if shift else None
        dqkv, dbias_ws, _ = o.window_attn_bwd(qkv, pw_b, geom.win2tok, Hp * Wp, dao, ao, lse, table, w, regions, geom.nW, geom.N, nH, scale)
        dtable = o.relpos_bias_bwd(dbias_ws, index, geom.N, table.shape[0]) if has_table else None
        dWpw, dbpw = _side_run(lambda: o.linear_wgrad(dqkv, bnout, want_bias=True), dqkv, bnout)
        dbn = o.linear_dgrad(dqkv, Wpw)
        # BatchNorm backward: d(d) = gamma rstd (dy - mean(dy) - xhat mean(dy xhat)), xhat = (d - mean) rstd
        red = o.bn_bwd_local(o.col_sums2(dbn, d), coef)   # (sum dy, sum dy*xhat) of this rank = (d beta, d gamma)
        dbet, dgam = red[0].clone(), red[1].clone()
        if eval_bn:  # fixed statistics: the normalisation is a per-channel affine map, no batch terms in its gradient
            abc = o.bn_bwd_coeffs(None, n, gam, coef)
        else:
            _allreduce_stats(red, group)
            abc = o.bn_bwd_coeffs(red, n, gam, coef)
        dd = o.col_affine2(dbn, abc[0], abc[2], d, abc[1])
        ddw = _side_run(lambda: o.dwconv3x3_wgrad(xp, dd, nB, Hp, Wp), xp, dd).view(C, 1, 3, 3)
        dxn = _crop_tokens(o.dwconv3x3(dd, dw9, nB, Hp, Wp, flip=True), nB, H, W, Hp, Wp)
        gx, dg1, db1 = o.layernorm_bwd(dxn, x.view(M, C), mean1, rstd1, g1, g_in=gy)
        _side_join()
        return (gx.view(nB, L, C), None, None, None, None, None, None, dg1, db1, ddw, dgam, dbet, dWpw.view(3 * C, C, 1, 1), dbpw,
                dWproj.view(C, C, 1, 1), dbproj, dtable, None, None)


_ZTAB = {}


def _zero_table(w, nH, device):
    key = (w, nH, str(device))
    t = _ZTAB.get(key)
    if t is None:
        t = torch.zeros(((2 * w - 1) ** 2, nH), dtype=torch.float32, device=device)
        _ZTAB[key] = t
    return t


class CvtFfnFn(torch.autograd.Function):
    """x + DropPath(FeedForward(LayerNorm(x)))  (cvt_v4_transformer.py:62-72): 1x1 conv -> QuickGELU -> 1x1 conv"""

    @staticmethod
    def forward(ctx, x, dp, g2, b2, W1p, bf1, W2p, bf2):
        o = ops_module()
        nB, L, C = x.shape
        x = x.contiguous()
        x2d = x.view(nB * L, C)
        Hd = W1p.shape[0]
        W1, W2 = _weight(W1p, (Hd, C)), _weight(W2p, (C, Hd))
        h, _, mean, rstd = o.layernorm_fwd(x2d, g2, b2, CVT_LN_EPS)
        a1g, a1 = o.linear_fwd(h, W1, bf1, gelu=True, want_preact=True, quick=True)
        y = o.linear_fwd(a1g, W2, bf2, residual=x2d, rowscale=dp, rows_per_sample=L, out_f32=True)
        ctx.dp = dp
        ctx.save_for_backward(x, mean, rstd, g2, h, a1, a1g, W1, W2)
        return y.view(nB, L, C)

    @staticmethod
    def backward(ctx, gy):
        o = ops_module()
        x, mean, rstd, g2, h, a1, a1g, W1, W2 = ctx.saved_tensors
        nB, L, C = x.shape
        M = nB * L
        Hd = W1.shape[0]
        gy = gy.contiguous().view(M, C)
        dyb = o.gather_cast(gy, M, rowscale=ctx.dp, rows_per_sample=L)
        dW2, dbf2 = _side_run(lambda: o.linear_wgrad(dyb, a1g, want_bias=True), dyb, a1g)
        da1 = o.linear_dgrad(dyb, W2, gelu_preact=a1, quick=True)
        dW1, dbf1 = _side_run(lambda: o.linear_wgrad(da1, h, want_bias=True), da1, h)
        dh = o.linear_dgrad(da1, W1)
        gx, dg2, db2 = o.layernorm_bwd(dh, x.view(M, C), mean, rstd, g2, g_in=gy)
        _side_join()
        return gx.view(nB, L, C), None, dg2, db2, dW1.view(Hd, C, 1, 1), dbf1, dW2.view(C, Hd, 1, 1), dbf2


# ------------------------------------------------------------------------------------------------
# monolithic ViT (models/vision_transformer.py:96-381): patch embedding without a norm, blocks with global attention
# ------------------------------------------------------------------------------------------------
class VitPatchEmbedFn(torch.autograd.Function):
    """PatchEmbed (vision_transformer.py:121-139): Conv2d(k = stride = patch) as im2col + GEMM, fp32 tokens out"""

    @staticmethod
    def forward(ctx, img, Wp, bp, patch):
        o = ops_module()
        E = Wp.shape[0]
        Kc = Wp.shape[1] * patch * patch
        nB, _, S, _ = img.shape
        cols = o.patch_im2col(img.contiguous(), patch, Kc)
        y = o.linear_fwd(cols, _weight(Wp, (E, Kc)), bp, out_f32=True)
        ctx.save_for_backward(cols)
        ctx.wparam, ctx.bparam, ctx.wshape = Wp, bp, tuple(Wp.shape)
        return y.view(nB, (S // patch) ** 2, E)

    @staticmethod
    def backward(ctx, gx):
        o = ops_module()
        (cols,) = ctx.saved_tensors
        M = cols.shape[0]
        dyb = o.gather_cast(gx.contiguous().view(M, -1), M)
        # (no bucket slots here: the embedding runs once per resolution group, so its parameters receive two contributions per
        # backward -- autograd sums them and the reducer's hook packs the sum)
        dW, dbp = o.linear_wgrad(dyb, cols, want_bias=True)
        return None, dW.view(ctx.wshape), dbp, None


VIT_WINDOW_TOKENS = 224  # crops of at most this many tokens run through the fused windowed kernels (one "window" per image)
_VIT_WIN = {}


def _vit_window(N, nH, device):
    """the geometry a crop of N tokens presents to the windowed kernels: one window per image holding tokens 0..N-1, no bias (a zero
    table over the smallest grid that has N positions -- the kernels mask the slots beyond N), no shift mask"""
    key = (N, nH, str(device))
    g = _VIT_WIN.get(key)
    if g is None:
        ws = 1
        while ws * ws < N:
            ws += 1
        g = (torch.arange(N, dtype=torch.int32, device=device), ws, _zero_table(ws, nH, device))
        _VIT_WIN[key] = g
    return g


def _vit_fused_route(N, hd, dtype):
    """window_attn.hip takes <= 64 tokens at head_dim 32 / 64; window_attn_big.hip <= 224 tokens at head_dim 32, or 64 in bf16"""
    if N <= 64:
        return hd in (32, 64)
    return N <= VIT_WINDOW_TOKENS and (hd == 32 or (hd == 64 and dtype == torch.bfloat16))


def vit_attention(o, qkv, bqkv, nB, N, nH, scale, save, out=None):
    """Attention.forward between the projections (vision_transformer.py:76-83) -> (out, tensors for vit_attention_bwd).  A crop is ONE
    window of the fused MFMA kernels: the 37 tokens of a 96^2 crop in the 64-slot kernels of window_attn.hip, the 197 tokens of a
    224^2 crop in the 224-slot flash-style kernels of window_attn_big.hip (head_dim 64 in bf16) -- no score matrix in HBM.  What
    does not fit (fp32 parity mode at head_dim 64, longer sequences) takes the batched-GEMM route of ops.vit_attn_fwd."""
    hd = qkv.shape[1] // 3 // nH
    if _vit_fused_route(N, hd, qkv.dtype):
        win2tok, ws, table = _vit_window(N, nH, qkv.device)
        frag = o.new_bias_frag(nH, N, qkv.device) if save else None
        ao, lse = o.window_attn_fwd(qkv, bqkv, win2tok, N, table, ws, None, 1, N, nH, scale, out=out, bias_frag=frag)
        if not save:
            return ao, ()
        return ao, ((qkv, ao, frag) if lse is None else (qkv, ao, frag, lse))
    ao, att = o.vit_attn_fwd(qkv, nB, N, nH, scale)
    if out is not None:
        out.copy_(ao)
        ao = out
    return ao, att


def vit_attention_bwd(o, dao, att, bqkv, nB, N, nH, scale, dqkv_out=None):
    if len(att) >= 3:  # the windowed routes: (qkv, out, bias fragments[, log-sum-exp of the 224-slot kernels])
        qkv, ao, frag = att[:3]
        lse = att[3] if len(att) == 4 else None
        win2tok, ws, _ = _vit_window(N, nH, qkv.device)
        return o.window_attn_bwd(qkv, bqkv, win2tok, N, dao, ao, lse, None, ws, None, 1, N, nH, scale, dqkv_out=dqkv_out, bias_frag=frag)[0]
    dqkv = o.vit_attn_bwd(dao, att, nB, N, nH, scale)
    if dqkv_out is not None:
        dqkv_out.copy_(dqkv)
        dqkv = dqkv_out
    return dqkv


def _vit_block_forward(x, nH, dp, prm, wts, save):
    """Block.forward (vision_transformer.py:110-116) on x fp32 [nB, N, C]"""
    o = ops_module()
    (g1, b1, bqkv, bproj, g2, b2, bfc1, bfc2) = prm
    (Wqkv, Wproj, W1, W2) = wts
    nB, N, C = x.shape
    x2d = x.view(nB * N, C)
    scale = (C // nH) ** -0.5
    dp1, dp2 = (None, None) if dp is None else dp
    xw, _, mean1, rstd1 = o.layernorm_fwd(x2d, g1, b1, LN_EPS)
    qkv = o.linear_fwd(xw, Wqkv, bqkv)
    ao, att = vit_attention(o, qkv, bqkv, nB, N, nH, scale, save)
    x1 = o.linear_fwd(ao, Wproj, bproj, residual=x2d, rowscale=dp1, rows_per_sample=N, out_f32=True)
    if not save and C in (192, 384) and _mlp_fused_infer(W1, C):
        # inference pass (the EMA teacher, the eval consumers): the branch in one kernel (deit_tiny / deit_small widths, whose fused kernels
        # take the plain weight cast)
        x2 = o.mlp_fused_fwd(x1, g2, b2, LN_EPS, W1, bfc1, W2, bfc2, rowscale=None if dp2 is None else dp2.repeat_interleave(N))
        return x2.view(nB, N, C), None, att
    h, _, mean2, rstd2 = o.layernorm_fwd(x1, g2, b2, LN_EPS)
    if save:
        a1g, a1 = o.linear_fwd(h, W1, bfc1, gelu=True, want_preact=True)
    else:
        a1g, a1 = o.linear_fwd(h, W1, bfc1, gelu=True), None
    x2 = o.linear_fwd(a1g, W2, bfc2, residual=x1, rowscale=dp2, rows_per_sample=N, out_f32=True)
    saved = (mean1, rstd1, xw, ao, x1, mean2, rstd2, h, a1, a1g) if save else None
    return x2.view(nB, N, C), saved, att


class VitBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, nH, dp, g1, b1, Wqkv_p, bqkv, Wproj_p, bproj, g2, b2, W1_p, bfc1, W2_p, bfc2):
        wts = (_weight(Wqkv_p), _weight(Wproj_p), _weight(W1_p), _weight(W2_p))
        x = x.contiguous()
        y, saved, att = _vit_block_forward(x, nH, dp, (g1, b1, bqkv, bproj, g2, b2, bfc1, bfc2), wts, True)
        ctx.nH, ctx.dp = nH, dp
        ctx.wparams = (Wqkv_p, Wproj_p, W1_p, W2_p)
        ctx.bparams = (bqkv, bproj, bfc1, bfc2)
        ctx.nparams = (g1, b1, g2, b2)
        ctx.natt = len(att)
        ctx.save_for_backward(x, g1, g2, bqkv, *wts, *saved, *att)
        return y

    @staticmethod
    def backward(ctx, gy):
        o = ops_module()
        nH, dp = ctx.nH, ctx.dp
        t = ctx.saved_tensors
        x, g1, g2, bqkv, Wqkv, Wproj, W1, W2, mean1, rstd1, xw, ao, x1, mean2, rstd2, h, a1, a1g = t[:18]
        att = tuple(t[18:18 + ctx.natt])
        nB, N, C = x.shape
        M = nB * N
        scale = (C // nH) ** -0.5
        dp1, dp2 = (None, None) if dp is None else dp
        gy = gy.contiguous().view(M, C)
        Wqkv_p, Wproj_p, W1_p, W2_p = ctx.wparams
        bqkv_p, bproj_p, bfc1_p, bfc2_p = ctx.bparams
        g1_p, b1_p, g2_p, b2_p = ctx.nparams
        dyb = o.gather_cast(gy, M, rowscale=dp2, rows_per_sample=N)
        dW2, dbfc2 = _wgrad(dyb, a1g, W2_p, want_bias=True, bias_param=bfc2_p)
        da1 = o.linear_dgrad(dyb, W2, gelu_preact=a1)
        dW1, dbfc1 = _wgrad(da1, h, W1_p, want_bias=True, bias_param=bfc1_p)
        dh = o.linear_dgrad(da1, W1)
        sink1, sink2 = _ln_sinks(g1_p, b1_p), _ln_sinks(g2_p, b2_p)
        gx1, dyw, dg2, db2 = o.layernorm_bwd_cast(dh, x1, mean2, rstd2, g2, g_in=gy, rowscale=dp1, rows_per_sample=N, gb_out=sink2)
        dWproj, dbproj = _wgrad(dyw, ao, Wproj_p, want_bias=True, bias_param=bproj_p)
        dao = o.linear_dgrad(dyw, Wproj)
        dqkv = vit_attention_bwd(o, dao, att, bqkv, nB, N, nH, scale)
        dWqkv, dbqkv = _wgrad(dqkv, xw, Wqkv_p, want_bias=True, bias_param=bqkv_p)
        dxw = o.linear_dgrad(dqkv, Wqkv)
        gx, dg1, db1 = o.layernorm_bwd(dxw, x.view(M, C), mean1, rstd1, g1, g_in=gx1, gb_out=sink1)
        return (gx.view(nB, N, C), None, None, _alias(dg1, sink1), _alias(db1, sink1), dWqkv, dbqkv, dWproj, dbproj,
                _alias(dg2, sink2), _alias(db2, sink2), dW1, dbfc1, dW2, dbfc2)


def vit_block(x, nH, dp, prm_list):
    if not torch.is_grad_enabled() or not (x.requires_grad or any(p.requires_grad for p in prm_list)):
        g1, b1, Wqkv_p, bqkv, Wproj_p, bproj, g2, b2, W1_p, bfc1, W2_p, bfc2 = prm_list
        wts = (_weight(Wqkv_p), _weight(Wproj_p), _weight(W1_p), _weight(W2_p))
        return _vit_block_forward(x.contiguous(), nH, dp, (g1, b1, bqkv, bproj, g2, b2, bfc1, bfc2), wts, False)[0]
    return VitBlockFn.apply(x, nH, dp, *prm_list)


def vit_block_attention(x, nH, prm_list):
    """the attention probabilities [nB, nH, N, N] of a block (Block.forward(return_attention=True), vision_transformer.py:112)"""
    o = ops_module()
    g1, b1, Wqkv_p, bqkv = prm_list[:4]
    nB, N, C = x.shape
    xw = o.layernorm_fwd(x.contiguous().view(nB * N, C), g1, b1, LN_EPS)[0]
    qkv = o.linear_fwd(xw, _weight(Wqkv_p), bqkv)
    _, att = o.vit_attn_fwd(qkv, nB, N, nH, (C // nH) ** -0.5)  # (evaluation hook: the batched-GEMM route keeps P in memory)
    p = att[-1]
    return p.reshape(nB, nH, p.shape[-2], p.shape[-1])[:, :, :N, :N].float()


# ------------------------------------------------------------------------------------------------
# Vision Longformer (models/vision_longformer.py:406-770): an AttnBlock followed by its MlpBlock is one autograd node, like a ViT
# block.  'full' stages project with one qkv Linear (vision_longformer.py:36-118); 'longformerhand' stages (layers/longformer2d.py)
# with `query` and `kv` Linears -- shared by local and global tokens (sharew) -- and restrict every local query to the global tokens
# plus its own and the eight adjacent w x w chunks (`chunk`, esvit_softmax_rows_chunked_fwd).  rpe off (the ape = 1 default of
# the reference's yaml files): no bias tables.
# ------------------------------------------------------------------------------------------------
def _vil_block_forward(x, nH, dp, chunk, prm, wts, save):
    o = ops_module()
    (g1, b1, bq, bkv, bproj, g2, b2, bfc1, bfc2) = prm
    (Wq, Wkv, Wproj, W1, W2) = wts
    nB, N, C = x.shape
    x2d = x.view(nB * N, C)
    scale = (C // nH) ** -0.5
    dp1, dp2 = (None, None) if dp is None else dp
    xw, _, mean1, rstd1 = o.layernorm_fwd(x2d, g1, b1, LN_EPS)
    if Wkv is None:   # one qkv Linear
        qkv, bqkv = o.linear_fwd(xw, Wq, bq), bq
    else:             # query | kv Linears: the same [q | k | v] column layout (longformer2d.py:160-162)
        qkv = torch.cat((o.linear_fwd(xw, Wq, bq), o.linear_fwd(xw, Wkv, bkv)), dim=1)
        bqkv = torch.cat((bq.detach(), bkv.detach()))
    if chunk is None:
        ao, att = vit_attention(o, qkv, bqkv, nB, N, nH, scale, save)
    else:
        ao, att = o.vit_attn_fwd(qkv, nB, N, nH, scale, chunk=chunk)
        att = att if save else ()
    x1 = o.linear_fwd(ao, Wproj, bproj, residual=x2d, rowscale=dp1, rows_per_sample=N, out_f32=True)
    h, _, mean2, rstd2 = o.layernorm_fwd(x1, g2, b2, LN_EPS)
    if save:
        a1g, a1 = o.linear_fwd(h, W1, bfc1, gelu=True, want_preact=True)
    else:
        a1g, a1 = o.linear_fwd(h, W1, bfc1, gelu=True), None
    x2 = o.linear_fwd(a1g, W2, bfc2, residual=x1, rowscale=dp2, rows_per_sample=N, out_f32=True)
    saved = (mean1, rstd1, xw, ao, x1, mean2, rstd2, h, a1, a1g, bqkv) if save else None
    return x2.view(nB, N, C), saved, att


class VilBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, nH, dp, chunk, g1, b1, Wq_p, bq, Wkv_p, bkv, Wproj_p, bproj, g2, b2, W1_p, bfc1, W2_p, bfc2):
        split = Wkv_p is not None
        wts = (_weight(Wq_p), _weight(Wkv_p) if split else None, _weight(Wproj_p), _weight(W1_p), _weight(W2_p))
        x = x.contiguous()
        y, saved, att = _vil_block_forward(x, nH, dp, chunk, (g1, b1, bq, bkv, bproj, g2, b2, bfc1, bfc2), wts, True)
        ctx.nH, ctx.dp, ctx.split, ctx.chunk = nH, dp, split, chunk
        ctx.wparams = (Wq_p, Wkv_p, Wproj_p, W1_p, W2_p)
        ctx.bparams = (bq, bkv, bproj, bfc1, bfc2)
        ctx.nparams = (g1, b1, g2, b2)
        ctx.natt = len(att)
        wsave = [w for w in wts if w is not None]
        ctx.save_for_backward(x, g1, g2, *wsave, *saved, *att)
        return y

    @staticmethod
    def backward(ctx, gy):
        o = ops_module()
        nH, dp, split = ctx.nH, ctx.dp, ctx.split
        t = list(ctx.saved_tensors)
        x, g1, g2 = t[:3]
        nw = 5 if split else 4
        ws = t[3:3 + nw]
        if split:
            Wq, Wkv, Wproj, W1, W2 = ws
        else:
            (Wq, Wproj, W1, W2), Wkv = ws, None
        mean1, rstd1, xw, ao, x1, mean2, rstd2, h, a1, a1g, bqkv = t[3 + nw:14 + nw]
        att = tuple(t[14 + nw:14 + nw + ctx.natt])
        nB, N, C = x.shape
        M = nB * N
        scale = (C // nH) ** -0.5
        dp1, dp2 = (None, None) if dp is None else dp
        gy = gy.contiguous().view(M, C)
        Wq_p, Wkv_p, Wproj_p, W1_p, W2_p = ctx.wparams
        bq_p, bkv_p, bproj_p, bfc1_p, bfc2_p = ctx.bparams
        g1_p, b1_p, g2_p, b2_p = ctx.nparams
        dyb = o.gather_cast(gy, M, rowscale=dp2, rows_per_sample=N)
        dW2, dbfc2 = _side_run(lambda: _wgrad(dyb, a1g, W2_p, want_bias=True, bias_param=bfc2_p), dyb, a1g)
        da1 = o.linear_dgrad(dyb, W2, gelu_preact=a1)
        dW1, dbfc1 = _side_run(lambda: _wgrad(da1, h, W1_p, want_bias=True, bias_param=bfc1_p), da1, h)
        dh = o.linear_dgrad(da1, W1)
        sink1, sink2 = _ln_sinks(g1_p, b1_p), _ln_sinks(g2_p, b2_p)
        gx1, dyw, dg2, db2 = o.layernorm_bwd_cast(dh, x1, mean2, rstd2, g2, g_in=gy, rowscale=dp1, rows_per_sample=N, gb_out=sink2)
        dWproj, dbproj = _side_run(lambda: _wgrad(dyw, ao, Wproj_p, want_bias=True, bias_param=bproj_p), dyw, ao)
        dao = o.linear_dgrad(dyw, Wproj)
        if ctx.chunk is not None:
            dqkv = o.vit_attn_bwd(dao, att, nB, N, nH, scale, chunk=ctx.chunk)
        else:
            dqkv = vit_attention_bwd(o, dao, att, bqkv, nB, N, nH, scale)
        if split:
            dq, dkv = dqkv[:, :C].contiguous(), dqkv[:, C:].contiguous()
            dWq, dbq = _side_run(lambda: _wgrad(dq, xw, Wq_p, want_bias=True, bias_param=bq_p), dq, xw)
            dWkv, dbkv = _side_run(lambda: _wgrad(dkv, xw, Wkv_p, want_bias=True, bias_param=bkv_p), dkv, xw)
            dxw = o.linear_dgrad(dqkv, torch.cat((Wq, Wkv), 0))  # dq Wq + dkv Wkv as one product over the stacked (tiny) weights
        else:
            dWq, dbq = _side_run(lambda: _wgrad(dqkv, xw, Wq_p, want_bias=True, bias_param=bq_p), dqkv, xw)
            dWkv, dbkv = None, None
            dxw = o.linear_dgrad(dqkv, Wq)
        gx, dg1, db1 = o.layernorm_bwd(dxw, x.view(M, C), mean1, rstd1, g1, g_in=gx1, gb_out=sink1)
        _side_join()
        return (gx.view(nB, N, C), None, None, None, _alias(dg1, sink1), _alias(db1, sink1), dWq, dbq, dWkv, dbkv, dWproj, dbproj,
                _alias(dg2, sink2), _alias(db2, sink2), dW1, dbfc1, dW2, dbfc2)


def vil_block(x, nH, dp, chunk, prm_list):
    """prm_list = (norm.w, norm.b, Wq | Wqkv, bq | bqkv, Wkv | None, bkv | None, Wproj, bproj, norm2.w, norm2.b, W1, b1, W2, b2)"""
    if not torch.is_grad_enabled() or not (x.requires_grad or any(p is not None and p.requires_grad for p in prm_list)):
        g1, b1, Wq_p, bq, Wkv_p, bkv, Wproj_p, bproj, g2, b2, W1_p, bfc1, W2_p, bfc2 = prm_list
        wts = (_weight(Wq_p), _weight(Wkv_p) if Wkv_p is not None else None, _weight(Wproj_p), _weight(W1_p), _weight(W2_p))
        return _vil_block_forward(x.contiguous(), nH, dp, chunk, (g1, b1, bq, bkv, bproj, g2, b2, bfc1, bfc2), wts, False)[0]
    return VilBlockFn.apply(x, nH, dp, chunk, *prm_list)


# ---- ragged multi-crop ViT block: all crops of a step in ONE set of LayerNorm / GEMM launches (cf. SwinBlockMultiFn) ------------
# LayerNorm, the four GEMMs and the residual adds are row-wise, so the token rows of the 224^2 crops and of the 96^2 crops run
# through them together; only the attention depends on the crop's token count and is launched once per resolution group on its
# row range.  Compared with one pass per group (vision_transformer.py:186-233): half the launches, ONE gradient contribution per
# parameter (no accumulation adds; the data-parallel reducer can overlap), half the split-K partial traffic of the weight gradients.
def _vit_block_forward_multi(X, segs, nH, dp_rows, prm, wts, save):
    """X fp32 [M, C]; segs: tuple of (row0, nB, N); dp_rows: None or (per-row DropPath scale of the attention branch [M], MLP branch [M])"""
    o = ops_module()
    (g1, b1, bqkv, bproj, g2, b2, bfc1, bfc2) = prm
    (Wqkv, Wproj, W1, W2) = wts
    M, C = X.shape
    scale = (C // nH) ** -0.5
    dp1, dp2 = (None, None) if dp_rows is None else dp_rows
    xw, _, mean1, rstd1 = o.layernorm_fwd(X, g1, b1, LN_EPS)
    qkv = o.linear_fwd(xw, Wqkv, bqkv)
    ao = torch.empty((M, C), dtype=qkv.dtype, device=X.device)
    atts = []
    for (r0, nB, N) in segs:
        r1 = r0 + nB * N
        _, att = vit_attention(o, qkv[r0:r1], bqkv, nB, N, nH, scale, save, out=ao[r0:r1])
        atts.append(att)
    x1 = o.linear_fwd(ao, Wproj, bproj, residual=X, rowscale=dp1, rows_per_sample=1, out_f32=True)
    if not save and C in (192, 384) and _mlp_fused_infer(W1, C):
        return o.mlp_fused_fwd(x1, g2, b2, LN_EPS, W1, bfc1, W2, bfc2, rowscale=dp2), None, atts  # (inference pass: see _vit_block_forward)
    h, _, mean2, rstd2 = o.layernorm_fwd(x1, g2, b2, LN_EPS)
    if save:
        a1g, a1 = o.linear_fwd(h, W1, bfc1, gelu=True, want_preact=True)
    else:
        a1g, a1 = o.linear_fwd(h, W1, bfc1, gelu=True), None
    x2 = o.linear_fwd(a1g, W2, bfc2, residual=x1, rowscale=dp2, rows_per_sample=1, out_f32=True)
    saved = (mean1, rstd1, xw, ao, x1, mean2, rstd2, h, a1, a1g) if save else None
    return x2, saved, atts


class VitBlockMultiFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, segs, nH, dp_rows, g1, b1, Wqkv_p, bqkv, Wproj_p, bproj, g2, b2, W1_p, bfc1, W2_p, bfc2):
        wts = (_weight(Wqkv_p), _weight(Wproj_p), _weight(W1_p), _weight(W2_p))
        X = X.contiguous()
        y, saved, atts = _vit_block_forward_multi(X, segs, nH, dp_rows, (g1, b1, bqkv, bproj, g2, b2, bfc1, bfc2), wts, True)
        ctx.segs, ctx.nH, ctx.dp_rows = segs, nH, dp_rows
        ctx.wparams = (Wqkv_p, Wproj_p, W1_p, W2_p)
        ctx.bparams = (bqkv, bproj, bfc1, bfc2)
        ctx.nparams = (g1, b1, g2, b2)
        ctx.natt = [len(a) for a in atts]
        ctx.save_for_backward(X, g1, g2, bqkv, *wts, *saved, *[t for a in atts for t in a])
        return y

    @staticmethod
    def backward(ctx, gy):
        o = ops_module()
        segs, nH, dp_rows = ctx.segs, ctx.nH, ctx.dp_rows
        t = ctx.saved_tensors
        X, g1, g2, bqkv, Wqkv, Wproj, W1, W2, mean1, rstd1, xw, ao, x1, mean2, rstd2, h, a1, a1g = t[:18]
        flat, atts, at = t[18:], [], 0
        for n in ctx.natt:
            atts.append(tuple(flat[at:at + n]))
            at += n
        M, C = X.shape
        scale = (C // nH) ** -0.5
        dp1, dp2 = (None, None) if dp_rows is None else dp_rows
        gy = gy.contiguous()
        Wqkv_p, Wproj_p, W1_p, W2_p = ctx.wparams
        bqkv_p, bproj_p, bfc1_p, bfc2_p = ctx.bparams
        g1_p, b1_p, g2_p, b2_p = ctx.nparams
        dyb = o.gather_cast(gy, M, rowscale=dp2, rows_per_sample=1)
        dW2, dbfc2 = _side_run(lambda: _wgrad(dyb, a1g, W2_p, want_bias=True, bias_param=bfc2_p), dyb, a1g)
        da1 = o.linear_dgrad(dyb, W2, gelu_preact=a1)
        dW1, dbfc1 = _side_run(lambda: _wgrad(da1, h, W1_p, want_bias=True, bias_param=bfc1_p), da1, h)
        dh = o.linear_dgrad(da1, W1)
        sink1, sink2 = _ln_sinks(g1_p, b1_p), _ln_sinks(g2_p, b2_p)
        gx1, dyw, dg2, db2 = o.layernorm_bwd_cast(dh, x1, mean2, rstd2, g2, g_in=gy, rowscale=dp1, rows_per_sample=1, gb_out=sink2)
        dWproj, dbproj = _side_run(lambda: _wgrad(dyw, ao, Wproj_p, want_bias=True, bias_param=bproj_p), dyw, ao)
        dao = o.linear_dgrad(dyw, Wproj)
        dqkv = torch.empty((M, 3 * C), dtype=dao.dtype, device=dao.device)
        for (r0, nB, N), att in zip(segs, atts):
            r1 = r0 + nB * N
            vit_attention_bwd(o, dao[r0:r1], att, bqkv, nB, N, nH, scale, dqkv_out=dqkv[r0:r1])
        dWqkv, dbqkv = _side_run(lambda: _wgrad(dqkv, xw, Wqkv_p, want_bias=True, bias_param=bqkv_p), dqkv, xw)
        dxw = o.linear_dgrad(dqkv, Wqkv)
        gx, dg1, db1 = o.layernorm_bwd(dxw, X, mean1, rstd1, g1, g_in=gx1, gb_out=sink1)
        _side_join()
        return (gx, None, None, None, _alias(dg1, sink1), _alias(db1, sink1), dWqkv, dbqkv, dWproj, dbproj,
                _alias(dg2, sink2), _alias(db2, sink2), dW1, dbfc1, dW2, dbfc2)


def vit_block_multi(X, segs, nH, dp_rows, prm_list):
    """one ViT block over the token rows of several resolution groups; X fp32 [M, C]"""
    if torch.is_grad_enabled() and (X.requires_grad or any(p.requires_grad for p in prm_list)):
        return VitBlockMultiFn.apply(X, segs, nH, dp_rows, *prm_list)
    g1, b1, Wqkv_p, bqkv, Wproj_p, bproj, g2, b2, W1_p, bfc1, W2_p, bfc2 = prm_list
    wts = (_weight(Wqkv_p), _weight(Wproj_p), _weight(W1_p), _weight(W2_p))
    return _vit_block_forward_multi(X.contiguous(), segs, nH, dp_rows, (g1, b1, bqkv, bproj, g2, b2, bfc1, bfc2), wts, False)[0]
