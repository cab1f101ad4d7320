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
"""
import numpy as np
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


def _weight(p, shape2d=None):
    """activation-dtype copy of an fp32 parameter; frozen parameters (the EMA teacher) are recast on every
    use because in-place `.data` updates (main_esvit.py:590) are invisible to version counters."""
    if p.requires_grad or P.is_managed(p):
        return P.cached_cast(p, shape2d)
    src = p.detach() if shape2d is None else p.detach().reshape(shape2d)
    return ops_module().cast_to_act(src.contiguous())


# ------------------------------------------------------------------------------------------------
# Swin block
# ------------------------------------------------------------------------------------------------
def _block_forward(x, geom, nH, index, dp, prm, wts, save):
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
    xw, _, mean1, rstd1 = o.layernorm_fwd(x2d, g1, b1, LN_EPS)
    qkv = o.linear_fwd(xw, Wqkv, bqkv)
    ao, lse = o.window_attn_fwd(qkv, bqkv, geom.win2tok, L, table, geom.ws, geom.region_ids, geom.nW, geom.N, nH, scale)
    x1 = o.linear_fwd(ao, Wproj, bproj, residual=x2d, rowscale=dp1, rows_per_sample=L, out_f32=True)
    h, _, mean2, rstd2 = o.layernorm_fwd(x1, g2, b2, LN_EPS)
    if save:
        a1g, a1 = o.linear_fwd(h, W1, bfc1, gelu=True, want_preact=True)
    else:
        a1g, a1 = o.linear_fwd(h, W1, bfc1, gelu=True), None
    x2 = o.linear_fwd(a1g, W2, bfc2, residual=x1, rowscale=dp2, rows_per_sample=L, out_f32=True)
    saved = (mean1, rstd1, xw, qkv, ao, x1, mean2, rstd2, h, a1, a1g, lse) if save else None
    return x2.view(nB, L, C), saved


class SwinBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, geom, nH, index, dp, g1, b1, table, Wqkv_p, bqkv, Wproj_p, bproj, g2, b2, W1_p, bfc1, W2_p, bfc2):
        wts = (_weight(Wqkv_p), _weight(Wproj_p), _weight(W1_p), _weight(W2_p))
        x = x.contiguous()
        y, saved = _block_forward(x, geom, nH, index, dp, (g1, b1, table, bqkv, bproj, g2, b2, bfc1, bfc2), wts, True)
        ctx.geom, ctx.nH, ctx.dp = geom, nH, dp
        ctx.save_for_backward(x, index, g1, table, g2, bqkv, *wts, *saved)
        return y

    @staticmethod
    def backward(ctx, gy):
        o = ops_module()
        geom, nH, dp = ctx.geom, ctx.nH, ctx.dp
        (x, index, g1, table, g2, bqkv, Wqkv, Wproj, W1, W2, mean1, rstd1, xw, qkv, ao, x1, mean2, rstd2, h, a1,
         a1g, lse) = ctx.saved_tensors
        nB, L, C = x.shape
        M = nB * L
        scale = (C // nH) ** -0.5
        dp1, dp2 = (None, None) if dp is None else dp
        gy = gy.contiguous().view(M, C)
        # ---- MLP branch ----
        dyb = o.gather_cast(gy, M, rowscale=dp2, rows_per_sample=L)
        dW2, dbfc2 = o.linear_wgrad(dyb, a1g, want_bias=True)
        da1 = o.linear_dgrad(dyb, W2, gelu_preact=a1)
        dW1, dbfc1 = o.linear_wgrad(da1, h, want_bias=True)
        dh = o.linear_dgrad(da1, W1)
        gx1, dg2, db2 = o.layernorm_bwd(dh, x1, mean2, rstd2, g2, g_in=gy)
        # ---- attention branch ----
        dyw = o.gather_cast(gx1, M, rowscale=dp1, rows_per_sample=L)
        dWproj, dbproj = o.linear_wgrad(dyw, ao, want_bias=True)
        dao = o.linear_dgrad(dyw, Wproj)
        dqkv, dbias_ws, dpad_ws = o.window_attn_bwd(qkv, bqkv, geom.win2tok, L, dao, ao, lse, table, geom.ws, geom.region_ids,
                                                    geom.nW, geom.N, nH, scale)
        dtable = o.relpos_bias_bwd(dbias_ws, index, geom.N, table.shape[0])
        dWqkv, dbqkv = o.linear_wgrad(dqkv, xw, want_bias=True)
        o.colsum(dpad_ws, out=dbqkv[C:], accumulate=True)  # k/v bias gradient from the zero-pad slots
        dxw = o.linear_dgrad(dqkv, Wqkv)
        gx, dg1, db1 = o.layernorm_bwd(dxw, x.view(M, C), mean1, rstd1, g1, g_in=gx1)
        return (gx.view(nB, L, C), None, None, None, None, dg1, db1, dtable, dWqkv, dbqkv, dWproj, dbproj, dg2, db2, dW1, dbfc1,
                dW2, dbfc2)


def swin_block(x, geom, nH, index, dp, prm_list):
    """prm_list: [g1, b1, table, Wqkv, bqkv, Wproj, bproj, g2, b2, W1, bfc1, W2, bfc2] (fp32 parameters)."""
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in prm_list)):
        return SwinBlockFn.apply(x, geom, nH, index, dp, *prm_list)
    g1, b1, table, Wqkv, bqkv, Wproj, bproj, g2, b2, W1, bfc1, W2, bfc2 = prm_list
    wts = (_weight(Wqkv), _weight(Wproj), _weight(W1), _weight(W2))
    y, _ = _block_forward(x.contiguous(), geom, nH, index, dp, (g1, b1, table, bqkv, bproj, g2, b2, bfc1, bfc2), wts, False)
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
        x, _, mean, rstd = o.layernorm_fwd(y, g, b, LN_EPS, dtype=torch.float32)
        ctx.save_for_backward(cols, y, mean, rstd, g)
        ctx.wshape = tuple(Wp.shape)
        return x.view(nB, (S // patch) ** 2, E)

    @staticmethod
    def backward(ctx, gx):
        o = ops_module()
        cols, y, mean, rstd, g = ctx.saved_tensors
        M, E = y.shape
        dy, dg, db = o.layernorm_bwd(gx.contiguous().view(M, E), y, mean, rstd, g)
        dyb = o.gather_cast(dy, M)
        dW, dbp = o.linear_wgrad(dyb, cols, want_bias=True)
        dW = dW.view(ctx.wshape)
        return None, dW, dbp, dg, db, None


def patch_embed_nonorm(img, Wp, bp, patch):
    raise NotImplementedError("PATCH_NORM False is not on the hot path")


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
        ctx.hw = (H, W)
        return out.view(nB, L // 4, 2 * C)

    @staticmethod
    def backward(ctx, go):
        o = ops_module()
        x, y, mean, rstd, g, Wc = ctx.saved_tensors
        H, W = ctx.hw
        rows = y.shape[0]
        gb = o.gather_cast(go.contiguous().view(rows, -1), rows)
        dWr = o.linear_wgrad(gb, y)
        dy = o.linear_dgrad(gb, Wc)
        dx, dg, db = o.merge_ln_bwd(dy, x, mean, rstd, g, H, W)
        return dx, None, None, dg, db, dWr


class FinalNormFn(torch.autograd.Function):
    """x fp32 [nB, T, C] -> LayerNorm(x) fp32 (the region features the loss matches on stay fp32)."""

    @staticmethod
    def forward(ctx, x, g, b):
        o = ops_module()
        x = x.contiguous()
        y, _, mean, rstd = o.layernorm_fwd(x.view(-1, x.shape[-1]), g, b, LN_EPS, dtype=torch.float32)
        ctx.save_for_backward(x, mean, rstd, g)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, gy):
        o = ops_module()
        x, mean, rstd, g = ctx.saved_tensors
        C = x.shape[-1]
        dx, dg, db = o.layernorm_bwd(gy.contiguous().view(-1, C), x.view(-1, C), mean, rstd, g)
        return dx.view(x.shape), dg, db


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


def _head_forward(x, prm, save):
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
    logits = o.linear_fwd(z, w)
    return logits, (W1, W2, W3, xa, h1, h1g, h2, h2g, z, inv, w, winv)


class DinoHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W1p, b1, W2p, b2, W3p, b3, v, g):
        logits, saved = _head_forward(x, (W1p, b1, W2p, b2, W3p, b3, v, g), True)
        ctx.save_for_backward(v, g, *saved)
        ctx.need_dg = g.requires_grad
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        o = ops_module()
        v, g, W1, W2, W3, xa, h1, h1g, h2, h2g, z, inv, w, winv = ctx.saved_tensors
        dlogits = dlogits.contiguous()
        dz = o.linear_dgrad(dlogits, w)
        dw = o.linear_wgrad(dlogits, z)
        dv, dg = o.weightnorm_bwd(dw, v, g, winv, ctx.need_dg)
        dh3 = o.l2norm_bwd(dz, z, inv)
        dW3, db3 = o.linear_wgrad(dh3, h2g, want_bias=True)
        dh2 = o.linear_dgrad(dh3, W3, gelu_preact=h2)
        dW2, db2 = o.linear_wgrad(dh2, h1g, want_bias=True)
        dh1 = o.linear_dgrad(dh2, W2, gelu_preact=h1)
        dW1, db1 = o.linear_wgrad(dh1, xa, want_bias=True)
        dx = o.linear_dgrad(dh1, W1, out_f32=True)
        return dx, dW1, db1, dW2, db2, dW3, db3, dv, dg


def dino_head(x, prm):
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in prm)):
        return DinoHeadFn.apply(x, *prm)
    return _head_forward(x, prm, False)[0]
