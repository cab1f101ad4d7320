"""TEST INFRASTRUCTURE ONLY -- plain-PyTorch restatement of every op in esvit_amd/ops.py.

Same function names and signatures as ``esvit_amd.ops`` so that (a) the ``-m gpu`` tests can
compare each HIP kernel with an independent implementation on identical inputs, and (b) the
CPU tests can monkeypatch ``esvit_amd.ops`` with this module to check the host-side composition
(manual backward formulas, index maps, autograd plumbing) against the reference model without
a GPU.  Nothing under esvit_amd/ imports this file; only tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() may.

Everything is computed in fp32 (fp64 where cheap) from the *given* inputs; outputs are rounded
to the activation dtype only at the points where the HIP kernels store activations.
"""
import numpy as np
import torch
import torch.nn.functional as F

_ACT_DTYPE = torch.float32


def set_act_dtype(dt):
    global _ACT_DTYPE
    _ACT_DTYPE = dt


def act_dtype():
    return _ACT_DTYPE


# ---- host-side integer maps (independent restatement with torch ops, swin_transformer.py) ----
def relative_position_index(ws):
    # swin_transformer.py:100-109
    co = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (co[:, :, None] - co[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1).numpy().astype(np.int64)


def _partition(x, ws):
    # swin_transformer.py:49-51
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


def window_maps(H, W, ws, shift):
    # swin_transformer.py:286-309 applied to token ids
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    ids = torch.arange(H * W, dtype=torch.float32).view(1, H, W, 1) + 1.0  # 0 reserved for pad
    ids = F.pad(ids, (0, 0, 0, Wp - W, 0, Hp - H))
    if shift > 0:
        ids = torch.roll(ids, shifts=(-shift, -shift), dims=(1, 2))
    win = _partition(ids, ws).view(-1).to(torch.int64) - 1
    win2tok = win.numpy().astype(np.int32)
    tok2win = np.empty(H * W, dtype=np.int32)
    slots = np.nonzero(win2tok >= 0)[0]
    tok2win[win2tok[slots]] = slots.astype(np.int32)
    return win2tok, tok2win


def shift_mask(H, W, ws, shift):
    # swin_transformer.py:249-272
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    img = torch.zeros((1, Hp, Wp, 1))
    cnt = 0
    for h in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for w in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, h, w, :] = cnt
            cnt += 1
    mw = _partition(img, ws).view(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    m = m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)
    return m.numpy().astype(np.float32)


# ---- helpers -----------------------------------------------------------------------------
def _r(x, dt=None):
    return x.to(dt or _ACT_DTYPE)


def _gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x * 0.7071067811865476))


def _gelu_grad(x):
    return 0.5 * (1.0 + torch.erf(x * 0.7071067811865476)) + x * torch.exp(-0.5 * x * x) * 0.3989422804014327


def _qgelu(x):
    return x * torch.sigmoid(1.702 * x)


def _qgelu_grad(x):
    s = torch.sigmoid(1.702 * x)
    return s * (1.0 + 1.702 * x * (1.0 - s))


def workspace(n, device, slot=0):
    return torch.empty(int(n), dtype=torch.float32, device=device)


# ---- GEMM family -------------------------------------------------------------------------
def row_stats_supported(dt, M, N):
    return dt == torch.bfloat16 and M > 0 and M % 128 == 0 and N % 128 == 0


def linear_fwd(x, w, bias=None, *, gelu=False, want_preact=False, residual=None, rowmap=None, rowmap_tokens=0,
               out_rows=None, rowscale=None, rows_per_sample=0, out_f32=False, quick=False, row_stats=None):
    if row_stats is not None:  # the logits as stored (rounded) and their softmax statistics (vision_transformer.py:418 + main_esvit.py:629,694)
        inv_temp, cen = row_stats[:2]
        y = _r(x.float() @ w.float().t(), x.dtype)
        mx, lse = teacher_row_stats(y, torch.zeros(y.shape[1], device=y.device) if cen is None else cen, inv_temp)
        if len(row_stats) > 2 and row_stats[2]:  # + the batch sums of the stored logits per column (esvit_gemm_desc::colstat)
            mx.esvit_col_sums = y.float().sum(0)
        return y, mx, lse
    acc = x.float() @ w.float().t()
    if bias is not None:
        acc = acc + bias
    pre = None
    if gelu:
        pre = _r(acc, x.dtype)
        acc = _qgelu(acc) if quick else _gelu(acc)
    M = x.shape[0]
    if rowmap is not None:
        period = rowmap.numel()
        m = torch.arange(M, device=x.device)
        t = rowmap.long()[m % period]
        keep = t >= 0
        drow = (m // period) * rowmap_tokens + t
        acc, drow = acc[keep], drow[keep]
        rows = out_rows
    else:
        drow = torch.arange(M, device=x.device)
        rows = M if out_rows is None else out_rows
    if rowscale is not None:
        acc = acc * rowscale[drow // rows_per_sample].unsqueeze(1)
    if residual is not None:
        acc = acc + residual[drow]
    y = torch.zeros((rows, w.shape[0]), dtype=torch.float32, device=x.device)
    if residual is not None and rowmap is not None:
        y = y  # rows never written by the kernel are undefined; tests only compare written rows
    y[drow] = acc
    y = y if out_f32 else _r(y, x.dtype)
    return (y, pre) if gelu and want_preact else y


def linear_dgrad(dy, w, *, gelu_preact=None, out_f32=False, quick=False):
    dx = dy.float() @ w.float()
    if gelu_preact is not None:
        dx = dx * (_qgelu_grad(gelu_preact.float()) if quick else _gelu_grad(gelu_preact.float()))
    return dx if out_f32 else _r(dx, dy.dtype)


def linear_wgrad(dy, x, *, out=None, accumulate=False, want_bias=False, db_out=None):
    dw = dy.float().t() @ x.float()
    if out is not None:
        if accumulate:
            out += dw
        else:
            out.copy_(dw)
        dw = out
    if not want_bias:
        return dw
    db = dy.float().sum(0)
    if db_out is not None:
        db_out.copy_(db)
        db = db_out
    return dw, db


def batched_nt(a, b, out_ld):
    P, M, K = a.shape
    N = b.shape[1]
    out = torch.zeros((P, M, out_ld), dtype=torch.float32, device=a.device)
    out[:, :, :N] = torch.bmm(a, b.transpose(1, 2))
    return out


def colsum(x, *, out=None, accumulate=False):
    s = x.float().sum(0)
    if out is not None:
        if accumulate:
            out += s
        else:
            out.copy_(s)
        return out
    return s


# ---- fused attention branch (esvit_attn_branch_fwd): the unfused sequence it replaces, same rounding points -------------------
def attn_branch_supported(dt, Cc, nH, N, rows=0, windows=0):
    return dt == torch.bfloat16 and Cc in (96, 192) and Cc == 32 * nH and N <= 64


def cast_weight(w, transpose=False, perm32=False):
    """(the restatement keeps the natural channel order)"""
    return _r(w.float().t().contiguous() if transpose else w.float(), torch.bfloat16 if _ACT_DTYPE == torch.bfloat16 else torch.float32)


def attn_branch_fwd(x, gamma, beta, eps, Wqkv_p, bqkv, Wproj_p, bproj, win2tok, L, rel_table, ws, region_ids, nW, N, nH, scale, *,
                    rowscale=None, out=None, bias_frag=None, save=False):
    xw, _, mean, rstd = layernorm_fwd(x, gamma, beta, eps, dtype=Wqkv_p.dtype)
    qkv = linear_fwd(xw, Wqkv_p, bqkv)
    ao = window_attn_fwd(qkv, bqkv, win2tok, L, rel_table, ws, region_ids, nW, N, nH, scale, bias_frag=bias_frag)[0]
    y = linear_fwd(ao, Wproj_p, bproj, residual=x, rowscale=rowscale, rows_per_sample=1, out_f32=True)
    if out is not None:
        out.copy_(y)
        y = out
    if save is True:
        return y, (xw, mean, rstd, qkv, ao)
    if save:
        for dst, src in zip(save, (xw, mean, rstd, qkv, ao)):
            dst.copy_(src)
        return y, save
    return y


# ---- normalisation ---------------------------------------------------------------------------
def mlp_fused_supported(dt, Cc, backward=False):
    return dt == torch.bfloat16 and Cc in (96, 192)


def mlp_fused_fwd(x, gamma, beta, eps, W1, b1, W2, b2, *, rowscale=None, next_norm=None):
    """the unfused sequence the fused kernel replaces (same rounding points)"""
    h, _, _, _ = layernorm_fwd(x, gamma, beta, eps, dtype=W1.dtype)
    act = linear_fwd(h, W1, b1, gelu=True)
    y = linear_fwd(act, W2, b2, residual=x, rowscale=rowscale, rows_per_sample=1, out_f32=True)
    if next_norm is None:
        return y
    xw, _, mean, rstd = layernorm_fwd(y, next_norm[0], next_norm[1], eps, dtype=W1.dtype)
    return y, (xw, mean, rstd)


def mlp_fused_fwd_train(x, gamma, beta, eps, W1, b1, W2, b2, *, rowscale=None):
    """the unfused training sequence esvit_mlp_fused_fwd_train replaces: (y, a1, a1g, h, mean, rstd)"""
    h, _, mean, rstd = layernorm_fwd(x, gamma, beta, eps, dtype=W1.dtype)
    a1g, a1 = linear_fwd(h, W1, b1, gelu=True, want_preact=True)
    y = linear_fwd(a1g, W2, b2, residual=x, rowscale=rowscale, rows_per_sample=1, out_f32=True)
    return y, a1, a1g, h, mean, rstd


def mlp_fused_train_supported(dt, Cc, rows=0):
    return False  # (the restatement keeps the unfused sequence: tests call mlp_fused_fwd_train directly)


def cast_transpose(w):
    return _r(w.float().t().contiguous(), torch.bfloat16 if _ACT_DTYPE == torch.bfloat16 else torch.float32)


MLP_W1_FWD, MLP_W1_BWD, MLP_W1T_BWD, MLP_W2T_BWD = range(4)


def mlp_fused_weight(kind, w):
    """(the restatement keeps the natural channel order)"""
    return _r(w.float()) if kind in (MLP_W1_FWD, MLP_W1_BWD) else cast_transpose(w)


def mlp_fused_bwd(x, gy, gamma, beta, eps, W1, W2T, W1T, b1, *, rowscale_mlp=None, rowscale_out=None):
    """restatement of esvit_mlp_fused_bwd (same rounding points: LN(x), GELU(A), dA and the scaled dy are rounded to the
    activation dtype where the kernel feeds them to an MFMA or stores them)"""
    dt = W1.dtype
    xf = x.float()
    mean = xf.mean(1, keepdim=True)
    rstd = torch.rsqrt(((xf - mean) ** 2).mean(1, keepdim=True) + eps)
    xhat = (xf - mean) * rstd
    xn = _r(xhat * gamma + beta, dt).float()
    dy = gy.float() if rowscale_mlp is None else gy.float() * rowscale_mlp[:, None]
    dyb = _r(dy, dt).float()
    A = xn @ W1.float().t() + b1
    cdf = 0.5 * (1 + torch.erf(A * 0.7071067811865476))
    a1g = _r(A * cdf, dt)
    dgelu = cdf + A * torch.exp(-0.5 * A * A) * 0.3989422804014327
    da1 = _r((dyb @ W2T.float().t()) * dgelu, dt)
    dh = da1.float() @ W1T.float().t()
    g = dh * gamma
    dx = rstd * (g - g.mean(1, keepdim=True) - xhat * (g * xhat).mean(1, keepdim=True))
    gx = gy.float() + dx
    gxa = _r(gx if rowscale_out is None else gx * rowscale_out[:, None], dt)
    return gx, gxa, _r(xhat, dt), a1g, da1


def ln_fold_finish(G, db, W, gamma, beta, *, gb_out=None):
    dgamma = (W * G).sum(0)
    dbeta = db @ W
    G.copy_(G * gamma + db[:, None] * beta[None, :])
    if gb_out is not None and gb_out[0] is not None and gb_out[1] is not None:
        gb_out[0].view(-1).copy_(dgamma)
        gb_out[1].view(-1).copy_(dbeta)
        return G, gb_out[0].view(-1), gb_out[1].view(-1)
    return G, dgamma, dbeta


def layernorm_fwd(x, gamma, beta, eps, *, rowmap=None, period_out=0, out_rows=None, want_f32=False, dtype=None):
    C = x.shape[-1]
    x2 = x.reshape(-1, C).float()
    mean = x2.mean(1)
    var = ((x2 - mean[:, None]) ** 2).mean(1)
    rstd = torch.rsqrt(var + eps)
    y = (x2 - mean[:, None]) * rstd[:, None] * gamma + beta
    dt = dtype or _ACT_DTYPE
    if rowmap is not None:
        T = rowmap.numel()
        r = torch.arange(x2.shape[0], device=x.device)
        ro = (r // T) * period_out + rowmap.long()[r % T]
        out = torch.zeros((out_rows, C), dtype=dt, device=x.device)
        out[ro] = _r(y, dt)
    else:
        out = _r(y, dt)
    return out, (y if want_f32 else None), mean, rstd


def _to_gb(dg, db, gb_out):
    if gb_out is not None and gb_out[0] is not None and gb_out[1] is not None:
        gb_out[0].view(-1).copy_(dg)
        gb_out[1].view(-1).copy_(db)
        return gb_out[0].view(-1), gb_out[1].view(-1)
    return dg, db


def layernorm_bwd(dy, x, mean, rstd, gamma, *, g_in=None, rowmap=None, period_in=0, gb_out=None):
    C = x.shape[-1]
    x2 = x.reshape(-1, C).float()
    if rowmap is not None:
        T = rowmap.numel()
        r = torch.arange(x2.shape[0], device=x.device)
        ri = (r // T) * period_in + rowmap.long()[r % T]
        d = dy.float()[ri]
    else:
        d = dy.float().reshape(-1, C)
    xh = (x2 - mean[:, None]) * rstd[:, None]
    gd = d * gamma
    dx = (gd - gd.mean(1, keepdim=True) - xh * (gd * xh).mean(1, keepdim=True)) * rstd[:, None]
    if g_in is not None:
        dx = dx + g_in.reshape(-1, C)
    dg, db = _to_gb((d * xh).sum(0), d.sum(0), gb_out)
    return dx.reshape(x.shape), dg, db


def layernorm_bwd_cast(dy, x, mean, rstd, gamma, *, g_in=None, rowscale=None, rows_per_sample=0, gb_out=None):
    dx, dg, db = layernorm_bwd(dy, x, mean, rstd, gamma, g_in=g_in, gb_out=gb_out)
    C = x.shape[-1]
    d2 = dx.reshape(-1, C)
    if rowscale is not None:
        d2 = d2 * rowscale[torch.arange(d2.shape[0], device=x.device) // rows_per_sample].unsqueeze(1)
    return dx, _r(d2, dy.dtype), dg, db


def layernorm_bwd_to_act(dy, x, mean, rstd, gamma, *, gb_out=None):
    """fp32 upstream gradient in, only the activation-dtype dx out (esvit_amd/ops.py: the patch-embedding norm)"""
    dx, dg, db = layernorm_bwd(dy.float(), x, mean, rstd, gamma, gb_out=gb_out)
    return _r(dx.reshape(-1, x.shape[-1])), dg, db


def _merge_gather(x, H, W):
    nB, L, C = x.shape
    xg = x.view(nB, H, W, C)
    return torch.cat([xg[:, 0::2, 0::2], xg[:, 1::2, 0::2], xg[:, 0::2, 1::2], xg[:, 1::2, 1::2]], -1).reshape(-1, 4 * C)


def merge_ln_fwd(x, gamma, beta, eps, H, W, dtype=None, out=None):
    g = _merge_gather(x.float(), H, W)
    y, _, mean, rstd = layernorm_fwd(g, gamma, beta, eps, dtype=dtype)
    if out is not None:
        out[0].copy_(y), out[1].copy_(mean), out[2].copy_(rstd)
        return out
    return y, mean, rstd


def merge_ln_bwd(dy, x, mean, rstd, gamma, H, W, dx_out=None, gb_out=None, accumulate=False, act_out=None, rowscale=None):
    nB, L, C = x.shape
    g = _merge_gather(x.float(), H, W)
    dg, dgamma, dbeta = layernorm_bwd(dy, g, mean, rstd, gamma)
    if gb_out is not None:
        if accumulate:
            gb_out[0].add_(dgamma)
            gb_out[1].add_(dbeta)
        else:
            gb_out[0].copy_(dgamma)
            gb_out[1].copy_(dbeta)
        dgamma, dbeta = gb_out[0], gb_out[1]
    dg = dg.view(nB, H // 2, W // 2, 4 * C)
    dx = torch.zeros((nB, H, W, C), dtype=torch.float32, device=x.device)
    dx[:, 0::2, 0::2] = dg[..., 0:C]
    dx[:, 1::2, 0::2] = dg[..., C:2 * C]
    dx[:, 0::2, 1::2] = dg[..., 2 * C:3 * C]
    dx[:, 1::2, 1::2] = dg[..., 3 * C:]
    if act_out is not None:  # cast(rowscale * dx): the MLP-branch operand of the block whose dL/dy this is
        d2 = dx.view(nB * L, C)
        if rowscale is not None:
            d2 = d2 * rowscale.view(-1, 1)
        act_out.view(nB * L, C).copy_(_r(d2, act_out.dtype))
    if dx_out is not None:
        dx_out.view(nB, L, C).copy_(dx.view(nB, L, C))
        return dx_out, dgamma, dbeta
    return dx.view(nB, L, C), dgamma, dbeta


# ---- data movement ---------------------------------------------------------------------------
def gather_cast(src, rows, *, rowmap=None, tokens=0, rowscale=None, rows_per_sample=0, dtype=None):
    C = src.shape[-1]
    s2 = src.reshape(-1, C).float()
    r = torch.arange(rows, device=src.device)
    if rowmap is not None:
        period = rowmap.numel()
        t = rowmap.long()[r % period]
        sr = (r // period) * tokens + t.clamp(min=0)
        v = s2[sr]
        if rowscale is not None:
            v = v * rowscale[sr // rows_per_sample].unsqueeze(1)
        v = v * (t >= 0).unsqueeze(1)
    else:
        v = s2[r]
        if rowscale is not None:
            v = v * rowscale[r // rows_per_sample].unsqueeze(1)
    return _r(v, dtype)


def cast_to_act(x, dtype=None):
    return _r(x, dtype)


def cast_to_f32(x):
    return x.float()


def transpose_cast(w, dtype=None):
    return _r(w.t().contiguous(), dtype)


def patch_im2col(img, P, Kpad, dtype=None, out=None):
    nB, ch, S, _ = img.shape
    G = S // P
    cols = img.view(nB, ch, G, P, G, P).permute(0, 2, 4, 1, 3, 5).reshape(nB * G * G, ch * P * P)
    full = torch.zeros((nB * G * G, Kpad), dtype=torch.float32, device=img.device)
    full[:, :ch * P * P] = cols
    if out is not None:
        out.copy_(_r(full, dtype))
        return out
    return _r(full, dtype)


def token_mean_fwd(x, dtype=None):
    m = x.float().mean(1)
    return m, _r(m, dtype)


def token_mean_bwd(g_mean, g_tok, T):
    dx = (g_mean / T).unsqueeze(1).expand(-1, T, -1).clone()
    if g_tok is not None:
        dx = dx + g_tok.view(dx.shape)
    return dx


# ---- window attention ------------------------------------------------------------------------
def _nt(N):
    """MFMA tiles per window side: 4 (64 padded tokens) for 7x7 windows, 14 (224) for 14x14"""
    if N <= 64:
        return 4
    if N <= 224:
        return 14
    raise RuntimeError("unsupported window")


def attn_frag_elems(N):
    return _nt(N) ** 2 * 256


def _frag_qk(nt):
    e = np.arange(nt * nt * 256)
    r, lane, f = e & 3, (e >> 2) & 63, e >> 8
    c, g = lane & 15, lane >> 4
    q = 16 * (f % nt) + c
    key = 16 * (f // nt) + 4 * g + r
    return q, key


def _dense_from_frag(frag, N):
    nt = _nt(N)
    q, key = _frag_qk(nt)
    dense = torch.zeros(frag.shape[:-1] + (16 * nt, 16 * nt), dtype=torch.float32, device=frag.device)
    dense[..., torch.as_tensor(q), torch.as_tensor(key)] = frag
    return dense[..., :N, :N]


def _frag_from_dense(dense, pad_key_value=0.0):
    N = dense.shape[-1]
    nt = _nt(N)
    full = torch.zeros(dense.shape[:-2] + (16 * nt, 16 * nt), dtype=torch.float32, device=dense.device)
    full[..., :, N:] = pad_key_value
    full[..., :N, :N] = dense
    q, key = _frag_qk(nt)
    return full[..., torch.as_tensor(q), torch.as_tensor(key)].contiguous()


def relpos_bias_fwd(table, index, N):
    nH = table.shape[1]
    index = index[:N, :N].contiguous()  # (N < ws^2: the first N positions of the grid -- ViT crops through the windowed kernels)
    dense = table[index.view(-1)].view(N, N, nH).permute(2, 0, 1).contiguous()
    return _frag_from_dense(dense, -1.0e30)


def dense_to_frag(dense):
    return _frag_from_dense(dense, 0.0)


def _to_windows(x, fill, win2tok, L, nW, N):
    """token-ordered [nB*L, D] -> window-ordered [nB*nW*N, D]; zero-pad slots take the row `fill` ([D])"""
    nB = x.shape[0] // L
    w2t = win2tok.long().view(1, nW * N)
    idx = (torch.arange(nB, device=x.device).view(nB, 1) * L + w2t.clamp(min=0)).reshape(-1)
    xw = x[idx].clone()
    pad = (w2t < 0).expand(nB, -1).reshape(-1)
    xw[pad] = fill.to(x.dtype)
    return xw, pad


def shift_region_ids(H, W, ws, shift):
    """region labels per window slot, recovered from the reference-style mask construction (swin_transformer.py:249-267)"""
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    img = torch.zeros((1, Hp, Wp, 1))
    cnt = 0
    for h in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for w in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, h, w, :] = cnt
            cnt += 1
    return _partition(img, ws).view(-1).numpy().astype(np.int32)


def _attn_core(qkvw, bias_frag, region_ids, nW, N, nH, scale):
    rows, C3 = qkvw.shape
    C = C3 // 3
    hd = C // nH
    Bw = rows // N
    dt = qkvw.dtype
    x = qkvw.float().view(Bw, N, 3, nH, hd).permute(2, 0, 3, 1, 4)
    q = _r(x[0] * scale, dt).float()  # kernel rounds scale*q to the activation dtype in LDS
    k, v = x[1], x[2]
    s = q @ k.transpose(-2, -1) + _dense_from_frag(bias_frag, N).unsqueeze(0)
    if region_ids is not None:
        ids = region_ids.view(nW, N)
        m = torch.where(ids[:, :, None] == ids[:, None, :], 0.0, -100.0)  # [nW, N, N]
        s = (s.view(Bw // nW, nW, nH, N, N) + m.view(1, nW, 1, N, N)).view(Bw, nH, N, N)
    p = torch.softmax(s, -1)
    return q, k, v, p


def new_bias_frag(nH, N, device):
    """the restatement's "fragment buffer" simply remembers the table the first call was given"""
    ws = 1
    while ws * ws < N:  # the smallest grid with N positions (N = ws^2 for Swin / CvT windows, fewer for ViT crops)
        ws += 1
    return torch.zeros(((2 * ws - 1) ** 2, nH), dtype=torch.float32, device=device)


def _table_of(rel_table, bias_frag):
    if bias_frag is not None:
        if rel_table is not None:
            bias_frag.copy_(rel_table.detach())
        return bias_frag
    return rel_table


def window_attn_fwd(qkv, qkv_bias, win2tok, L, rel_table, ws, region_ids, nW, N, nH, scale, want_attn=False, out=None, bias_frag=None):
    out_arg = out
    rel_table = _table_of(rel_table, bias_frag)
    bias_frag = relpos_bias_fwd(rel_table, torch.as_tensor(relative_position_index(ws), device=qkv.device), N)
    C = qkv.shape[1] // 3
    qkvw, pad = _to_windows(qkv, _r(qkv_bias, qkv.dtype), win2tok, L, nW, N)
    q, k, v, p = _attn_core(qkvw, bias_frag, region_ids, nW, N, nH, scale)
    pr = _r(p, qkv.dtype).float()  # P is rounded to the activation dtype before P@V
    ow = (pr @ v).transpose(1, 2).reshape(qkvw.shape[0], C)
    nB = qkv.shape[0] // L
    idx = (torch.arange(nB, device=qkv.device).view(nB, 1) * L + win2tok.long().view(1, -1).clamp(min=0)).reshape(-1)
    out = torch.zeros((qkv.shape[0], C), dtype=torch.float32, device=qkv.device)
    out[idx[~pad]] = ow[~pad]
    if out_arg is not None:
        out_arg.copy_(_r(out, qkv.dtype))
        out = out_arg
    else:
        out = _r(out, qkv.dtype)
    return (out, None, p) if want_attn else (out, None)


def attn_dbias_slabs(N, windows, nH, device):
    """esvit_amd/ops.py: one buffer for the bias-gradient slabs of several window_attn_bwd calls (here: one slab per call)"""
    buf = torch.zeros((len(windows), nH, attn_frag_elems(N)), dtype=torch.float32, device=device)
    return buf, [buf[i:i + 1] for i in range(len(windows))]


def window_attn_bwd(qkv, qkv_bias, win2tok, L, dout, fwd_out, lse, rel_table, ws, region_ids, nW, N, nH, scale, dqkv_out=None, bias_frag=None,
                    dbias_out=None):
    rel_table = _table_of(rel_table, bias_frag)
    bias_frag = relpos_bias_fwd(rel_table, torch.as_tensor(relative_position_index(ws), device=qkv.device), N)
    C = qkv.shape[1] // 3
    dt = qkv.dtype
    qkvw, pad = _to_windows(qkv, _r(qkv_bias, dt), win2tok, L, nW, N)
    dow, _ = _to_windows(dout, torch.zeros(C, device=qkv.device), win2tok, L, nW, N)
    q, k, v, p = _attn_core(qkvw, bias_frag, region_ids, nW, N, nH, scale)
    Bw = qkvw.shape[0] // N
    hd = q.shape[-1]
    do = dow.float().view(Bw, N, nH, hd).permute(0, 2, 1, 3)
    pr = _r(p, dt).float()
    dv = pr.transpose(-2, -1) @ do
    dp = do @ v.transpose(-2, -1)
    ds = pr * (dp - (pr * dp).sum(-1, keepdim=True))  # the kernel re-reads P from its (activation-dtype) LDS image
    dsr = _r(ds, dt).float()
    dq = (dsr @ k) * scale
    dk = dsr.transpose(-2, -1) @ q
    dqkvw = torch.stack([dq, dk, dv], 0).permute(1, 3, 0, 2, 4).reshape(qkvw.shape)  # [Bw*N, 3C] fp32
    nB = qkv.shape[0] // L
    idx = (torch.arange(nB, device=qkv.device).view(nB, 1) * L + win2tok.long().view(1, -1).clamp(min=0)).reshape(-1)
    dqkv = torch.zeros(qkv.shape, dtype=torch.float32, device=qkv.device)
    dqkv[idx[~pad]] = dqkvw[~pad]
    dpad = dqkvw[pad][:, C:].sum(0, keepdim=True) if pad.any() else torch.zeros((1, 2 * C), device=qkv.device)
    dbias = ds.sum(0)  # [nH, N, N]
    ws = _frag_from_dense(dbias, 0.0).unsqueeze(0)  # parts = 1
    if dbias_out is not None:
        dbias_out.copy_(ws)
        ws = dbias_out
    if dqkv_out is not None:
        dqkv_out.copy_(_r(dqkv, dt))
        return dqkv_out, ws, dpad
    return _r(dqkv, dt), ws, dpad


def relpos_bias_bwd(dbias_ws, index, N, table_rows, out=None, accumulate=None):
    nH = dbias_ws.shape[1]
    dense = _dense_from_frag(dbias_ws.sum(0), N)  # [nH, N, N]
    dtable = torch.zeros((table_rows, nH), dtype=torch.float32, device=dbias_ws.device) if out is None else out
    if out is not None and accumulate is not None and not accumulate:
        dtable.zero_()
    dtable.index_add_(0, index.view(-1), dense.permute(1, 2, 0).reshape(N * N, nH))
    return dtable


# ---- DINOHead pieces ---------------------------------------------------------------------------
def l2norm_fwd(x):
    xf = x.float()
    inv = 1.0 / xf.norm(dim=1).clamp_min(1e-12)
    return _r(xf * inv[:, None], x.dtype), inv


def l2norm_bwd(dz, z, inv):
    g, zz = dz.float(), z.float()
    return _r((g - zz * (g * zz).sum(1, keepdim=True)) * inv[:, None], z.dtype)


def weightnorm_fwd(v, g, dtype=None):
    inv = 1.0 / v.norm(dim=1)
    return _r(v * (g.view(-1) * inv)[:, None], dtype), inv


def weightnorm_bwd(dw, v, g, inv, need_dg, dv_out=None):
    vh = v * inv[:, None]
    dot = (dw * vh).sum(1, keepdim=True)
    dv = (dw - vh * dot) * (g.view(-1, 1) * inv[:, None])
    if dv_out is not None:
        dv_out.copy_(dv)
        dv = dv_out
    return dv, (dot if need_dg else None)


# ---- loss ----------------------------------------------------------------------------------------
def teacher_row_stats(t, center, inv_temp):
    z = (t.float() - center.view(1, -1)) * inv_temp
    mx = z.max(1).values
    return mx, torch.log(torch.exp(z - mx[:, None]).sum(1))


def row_argmax(sim, Tt):
    ld = sim.shape[-1]
    return sim.reshape(-1, ld)[:, :Tt].argmax(1).to(torch.int32)


def region_match(sim, Tt, crop_id, cm_row, tmatch):
    B, S, ld = sim.shape
    for iq in range(2):
        j = sim[:, :, iq * Tt:(iq + 1) * Tt].argmax(-1)                       # [B, S]
        val = iq * B * Tt + torch.arange(B, device=sim.device)[:, None] * Tt + j
        val = torch.where(crop_id.view(1, S) == iq, torch.full_like(val, -1), val)
        tmatch[cm_row.long(), iq] = val.reshape(-1).to(torch.int32)
    return tmatch


def dino_ce(s, t, center, t_max, t_lse, tmatch, row_w, inv_student_temp, inv_teacher_temp, row_loss=None, term_w=None, row_order=None,
            s_stats=None):
    z = s.float() * inv_student_temp
    lse = torch.logsumexp(z, 1) if s_stats is None else s_stats[0] + s_stats[1]
    ps = torch.exp(z - lse[:, None])
    if term_w is not None:  # four individually weighted terms per row (mixup targets)
        tm = tmatch.view(-1, 4).long()
        w = term_w.view(-1, 4) * (tm >= 0)
        pt = torch.zeros_like(z)
        for j in range(4):
            ii = tm[:, j].clamp(min=0)
            pj = torch.exp((t.float()[ii] - center.view(1, -1)) * inv_teacher_temp - (t_max[ii] + t_lse[ii])[:, None])
            pt = pt + pj * w[:, j:j + 1]
        wsum = w.sum(1)
        rl = wsum * lse - (pt * z).sum(1)
        if row_loss is not None:
            row_loss.copy_(rl)
        else:
            row_loss = rl
        return row_loss, _r(inv_student_temp * (wsum[:, None] * ps - pt), s.dtype)
    tm = tmatch.view(-1, 2).long()
    pt = torch.zeros_like(z)
    nterms = (tm >= 0).sum(1).float()
    for j in range(2):
        idx = tm[:, j]
        ok = idx >= 0
        ii = idx.clamp(min=0)
        pj = torch.exp((t.float()[ii] - center.view(1, -1)) * inv_teacher_temp - (t_max[ii] + t_lse[ii])[:, None])
        pt = pt + pj * ok[:, None]
    rl = row_w * (nterms * lse - (pt * z).sum(1))
    if row_loss is not None:
        row_loss.copy_(rl)
    else:
        row_loss = rl
    ds = (row_w * inv_student_temp)[:, None] * (nterms[:, None] * ps - pt)
    return row_loss, _r(ds, s.dtype)


def sum_f32(x):
    return x.sum()


def scale_inplace(x, scale):
    x.mul_(scale.to(x.dtype) if x.dtype == torch.float32 else scale)
    return x


def center_ema(center, colsum_, momentum, denom):
    center.copy_(center * momentum + colsum_.view_as(center) / denom * (1 - momentum))
    return center


# ---- CvT backbone pieces ----------------------------------------------------------------------
def conv_out_size(n, k, stride, pad):
    return (n + 2 * pad - k) // stride + 1


def conv_im2col(src, nchw, nB, H, W, Cin, k, stride, pad, dtype=None):
    import torch.nn.functional as F
    dt = dtype or (_ACT_DTYPE if nchw else src.dtype)
    x = src.float() if nchw else src.float().view(nB, H, W, Cin).permute(0, 3, 1, 2)
    Ho, Wo = conv_out_size(H, k, stride, pad), conv_out_size(W, k, stride, pad)
    u = F.unfold(x, k, padding=pad, stride=stride)                      # [nB, Cin*k*k, Ho*Wo], channel-major (c, ky, kx)
    u = u.view(nB, Cin, k * k, Ho * Wo).permute(0, 3, 2, 1).reshape(nB * Ho * Wo, k * k * Cin)  # -> (ky, kx, c)
    Kpad = -(-(k * k * Cin) // 8) * 8
    cols = torch.zeros((nB * Ho * Wo, Kpad), dtype=torch.float32, device=src.device)
    cols[:, :k * k * Cin] = u
    return _r(cols, dt)


def conv_col2im(dcols, nB, H, W, Cin, k, stride, pad):
    import torch.nn.functional as F
    Ho, Wo = conv_out_size(H, k, stride, pad), conv_out_size(W, k, stride, pad)
    u = dcols.float()[:, :k * k * Cin].view(nB, Ho * Wo, k * k, Cin).permute(0, 3, 2, 1).reshape(nB, Cin * k * k, Ho * Wo)
    x = F.fold(u, (H, W), k, padding=pad, stride=stride)                # [nB, Cin, H, W]
    return x.permute(0, 2, 3, 1).reshape(nB * H * W, Cin).contiguous()


def dwconv3x3(x, w, nB, H, W, flip=False):
    import torch.nn.functional as F
    Cc = x.shape[1]
    wk = w.float().view(Cc, 1, 3, 3)
    if flip:
        wk = wk.flip(2, 3)
    y = F.conv2d(x.float().view(nB, H, W, Cc).permute(0, 3, 1, 2), wk, padding=1, groups=Cc)
    return _r(y.permute(0, 2, 3, 1).reshape(nB * H * W, Cc), x.dtype)


def dwconv3x3_wgrad(x, dy, nB, H, W):
    import torch.nn.functional as F
    Cc = x.shape[1]
    xp = F.pad(x.float().view(nB, H, W, Cc), (0, 0, 1, 1, 1, 1))
    g = dy.float().view(nB, H, W, Cc)
    dw = torch.stack([(xp[:, ky:ky + H, kx:kx + W] * g).sum((0, 1, 2)) for ky in range(3) for kx in range(3)], 1)
    return dw.contiguous()


def col_sums2(a, b):
    af = a.float()
    return torch.stack([af.sum(0), (af * b.float()).sum(0)], 0)


def col_affine2(x1, a1, a3, x2=None, a2=None, act=0):
    y = x1.float() * a1 + a3
    if act == 1:
        y = F.gelu(y)
    elif act == 2:
        y = x2.float() * (0.5 * (1.0 + torch.erf(y * 0.7071067811865476)) + y * torch.exp(-0.5 * y * y) * 0.3989422804014327)
    elif act == 3:
        y = y.clamp_min(0.0)
    elif act == 4:
        y = torch.where(y > 0, x2.float(), torch.zeros_like(y))
    elif x2 is not None:
        y = y + x2.float() * a2
    return _r(y, x1.dtype)


def bn_fwd_coeffs(sums, n, gamma, beta, eps, momentum, running_mean=None, running_var=None):
    n = float(n)
    mean = sums[0] / n
    var = (sums[1] / n - mean * mean).clamp_min(0.0)
    rstd = torch.rsqrt(var + eps)
    a = gamma * rstd
    if running_mean is not None:
        running_mean.mul_(1 - momentum).add_(momentum * mean)
        running_var.mul_(1 - momentum).add_(momentum * var * (n / (n - 1)))
    return torch.stack([a, beta - mean * a, mean, rstd], 0)


def bn_eval_coeffs(running_mean, running_var, gamma, beta, eps):
    rstd = torch.rsqrt(running_var + eps)
    a = gamma * rstd
    return torch.stack([a, beta - running_mean * a, running_mean.clone(), rstd], 0)


def bn_bwd_local(sums, coef):
    return torch.stack([sums[0], coef[3] * (sums[1] - coef[2] * sums[0])], 0)


def bn_bwd_coeffs(red, n, gamma, coef):
    mean, rstd = coef[2], coef[3]
    m1 = red[0] / float(n) if red is not None else torch.zeros_like(mean)
    m2 = red[1] / float(n) if red is not None else torch.zeros_like(mean)
    A = gamma * rstd
    B = -gamma * rstd * rstd * m2
    return torch.stack([A, B, -gamma * rstd * m1 - B * mean], 0)


def pad_crop_tokens(src, nB, Hs, Ws, Hd, Wd):
    Cc = src.shape[1]
    out = torch.zeros((nB, Hd, Wd, Cc), dtype=src.dtype, device=src.device)
    h, w = min(Hs, Hd), min(Ws, Wd)
    out[:, :h, :w] = src.view(nB, Hs, Ws, Cc)[:, :h, :w]
    return out.view(nB * Hd * Wd, Cc)


# ------------------------------------------------------------------------------------------------
# global attention of the monolithic ViT backbones (models/vision_transformer.py:67-94), unfused as the product runs it:
# S and P are materialised in the activation dtype, the softmax itself is fp32
# ------------------------------------------------------------------------------------------------
def vit_pad_tokens(N):
    return -(-N // 16) * 16


def _vit_heads(qkv, B, N, nH, parts):
    C = qkv.shape[1] // parts
    return qkv.view(B, N, parts, nH, C // nH).permute(2, 0, 3, 1, 4).float()  # [parts, B, nH, N, hd]


def chunk_mask(chunk):
    """[N, N] bool: may query i see key j?  (layers/slidingchunk_2d.py:268-287 exact = 0, layers/longformer2d.py:163-262, 310-327: global
    tokens see and are seen by everything, local tokens see their own and the eight adjacent w x w chunks)"""
    c = chunk.long()
    glob = c < 0
    cx, cy = c >> 16, c & 0xffff
    near = ((cx[:, None] - cx[None, :]).abs() <= 1) & ((cy[:, None] - cy[None, :]).abs() <= 1)
    return near | glob[:, None] | glob[None, :]


def vit_attn_fwd(qkv, B, N, nH, scale, chunk=None):
    dt = qkv.dtype
    q, k, v = _vit_heads(qkv, B, N, nH, 3)
    s = _r(q @ k.transpose(-2, -1), dt).float()
    if chunk is not None:
        s = s.masked_fill(~chunk_mask(chunk[0] if isinstance(chunk, tuple) else chunk).to(s.device), float("-inf"))
    p = _r(torch.softmax(scale * s, dim=-1), dt).float()
    o = _r(p @ v, dt)
    C = qkv.shape[1] // 3
    return o.transpose(1, 2).reshape(B * N, C).contiguous(), (torch.stack((q, k, v)), p)  # (two tensors, like ops.vit_attn_fwd)


def vit_attn_bwd(dout, saved, B, N, nH, scale, chunk=None):
    (q, k, v), p = saved
    dt = dout.dtype
    do = _vit_heads(dout, B, N, nH, 1)[0]
    dv = _r(p.transpose(-2, -1) @ do, dt)
    dp = _r(do @ v.transpose(-2, -1), dt).float()
    ds = _r(scale * p * (dp - (p * dp).sum(-1, keepdim=True)), dt).float()
    dq = _r(ds @ k, dt)
    dk = _r(ds.transpose(-2, -1) @ q, dt)
    C = dout.shape[1]
    return torch.stack([dq, dk, dv], 0).permute(1, 3, 0, 2, 4).reshape(B * N, 3 * C).to(dt).contiguous()
