"""MI355X-native Swin backbone behind the reference's module interface (models/swin_transformer.py).

The module tree only *holds parameters* -- names, shapes, dtypes, registration order and init are those of
the reference (so state_dicts, ``.parameters()`` zips for the EMA, DDP and ``get_params_groups`` behave
identically, SURVEY.md 5 "checkpoint / resume") -- while every forward runs the fused HIP pipeline in
esvit_amd.functional: one autograd node per Swin block, window partition / roll / pad folded into kernel
address maps, bf16 MFMA GEMMs with fused epilogues.
"""
import logging
import os
from functools import partial
from math import sqrt

import numpy as np
import torch
import torch.nn as nn

from .. import functional as Fn
from .registry import register_model


def _trunc_normal_(t, std=0.02):
    return nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)


def _pair(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


class DropPath(nn.Module):
    """holder for the stochastic-depth rate; the per-sample factors are drawn in the block (vision_transformer.py:30-38)"""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def factors(self, nB, device):
        if not self.drop_prob or not self.training:
            return None
        keep = 1.0 - self.drop_prob
        return (keep + torch.rand(nB, device=device)).floor_().div_(keep)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        assert drop == 0., "dropout is 0 in every reference yaml"
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop = nn.Dropout(drop)


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        assert qkv_bias and qk_scale is None and attn_drop == 0. and proj_drop == 0.
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        self.scale = (dim // num_heads) ** -0.5
        ws = window_size[0]
        if ws * ws > 64 and dim // num_heads != 32:
            # the head_dim-64 instances of the 224-slot kernels serve the bias-free ViT crops: they do not produce the gradient of
            # the relative-position table (every reference Swin configuration has head_dim 32)
            raise NotImplementedError("windows of more than 64 tokens need head_dim 32 (got %d)" % (dim // num_heads))
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), num_heads))
        p = np.arange(ws * ws)
        rel = (p[:, None] // ws - p[None, :] // ws + ws - 1) * (2 * ws - 1) + (p[:, None] % ws - p[None, :] % ws + ws - 1)
        self.register_buffer("relative_position_index", torch.from_numpy(rel.astype(np.int64)))
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        _trunc_normal_(self.relative_position_bias_table, std=.02)
        self.softmax = nn.Softmax(dim=-1)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size=7, shift_size=0, mlp_ratio=4., qkv_bias=True,
                 qk_scale=None, drop=0., attn_drop=0., drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, input_resolution, num_heads
        self.window_size, self.shift_size, self.mlp_ratio = window_size, shift_size, mlp_ratio
        if min(input_resolution) <= window_size:  # swin_transformer.py:206-209
            self.shift_size = 0
            self.window_size = min(input_resolution)
        assert 0 <= self.shift_size < self.window_size
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, _pair(self.window_size), num_heads, qkv_bias, qk_scale, attn_drop, drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.H, self.W = input_resolution

    def _params(self):
        a, m = self.attn, self.mlp
        return [self.norm1.weight, self.norm1.bias, a.relative_position_bias_table, a.qkv.weight, a.qkv.bias, a.proj.weight,
                a.proj.bias, self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias]

    def forward(self, x, return_attention=False):
        B, L, C = x.shape
        H = W = int(sqrt(L))
        geom = Fn.geometry(H, W, self.window_size, self.shift_size, x.device)
        dp = None
        if isinstance(self.drop_path, DropPath):
            dp = self.__dict__.pop("_dp_pending", None)  # drawn for every block at once by SwinTransformer._tokens
            if dp is None or dp[0].shape[0] != B:
                f1, f2 = self.drop_path.factors(B, x.device), self.drop_path.factors(B, x.device)
                dp = None if f1 is None else (f1, f2)
        attn = None
        if return_attention:
            attn = Fn.swin_block_attention(x, geom, self.num_heads, self.attn.relative_position_index, self._params())
        y = Fn.swin_block(x, geom, self.num_heads, self.attn.relative_position_index, dp, self._params())
        return y, attn


def _block_forward_multi(blk, X, groups, dp, shadow=None, prev_scale=None, pre=None, next_blk=None):
    """blk: SwinTransformerBlock; X fp32 [M, C]; groups: list of (row0, nB, H, W); dp: None or the per-ROW DropPath scales
    (attention branch [M], MLP branch [M]; rows of one sample share its factor).
    shadow / prev_scale: the previous block's shadow output and MLP-branch DropPath row scale (Fn.SwinBlockMultiFn).
    pre: norm1(X) + statistics if the previous block's fused MLP kernel produced them; next_blk: the next block of the stage (its
    norm1 then rides on this block's fused MLP kernel).
    -> (y, shadow of y, this block's MLP-branch row scale, the next block's pre or None)"""
    segs = tuple((r0, nB, H * W, Fn.geometry(H, W, blk.window_size, blk.shift_size, X.device)) for (r0, nB, H, W) in groups)
    dp_rows = dp
    nn_ = None if next_blk is None else (next_blk.norm1.weight, next_blk.norm1.bias)
    y, ysh, nxt = Fn.swin_block_multi(X, segs, blk.num_heads, blk.attn.relative_position_index, dp_rows, blk._params(), shadow, prev_scale,
                                      pre, nn_)
    return y, ysh, (None if dp_rows is None else dp_rows[1]), nxt


class PatchMerging(nn.Module):
    def __init__(self, input_resolution, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.input_resolution, self.dim = input_resolution, dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)

    def forward(self, x):
        B, L, C = x.shape
        H = W = int(sqrt(L))
        if H % 2 or W % 2:  # odd feature map: zero row / column appended first (swin_transformer.py:406-408)
            x = Fn.PadTokensFn.apply(x, H, W, H + H % 2, W + W % 2)
            H, W = H + H % 2, W + W % 2
        return Fn.PatchMergeFn.apply(x, H, W, self.norm.weight, self.norm.bias, self.reduction.weight)

    def forward_ragged(self, X, groups, shadow=None, prev_scale=None):
        """X fp32 [M, C] token rows of several resolution groups ((row0, nB, H, W) each) -> ([M/4, 2C], merged groups, shadow of the
        output).  shadow / prev_scale: the stage's last block's shadow output and MLP-branch DropPath row scale -- the merge's
        backward then emits that block's cast gradient; the returned shadow is for the next stage's first block, which hands this
        node the cast dL/dY its GEMMs read (Fn.SwinBlockMultiFn)"""
        if any(H % 2 or W % 2 for (_, _, H, W) in groups):
            raise NotImplementedError("odd feature maps on the ragged multi-crop route: set model.ragged_multi_crop = False "
                                      "(the per-group schedule pads them, swin_transformer.py:406-408)")
        Y, Ysh = Fn.PatchMergeMultiFn.apply(X, shadow, prev_scale, tuple(groups), self.norm.weight, self.norm.bias, self.reduction.weight)
        return Y, [(r0 // 4, nB, H // 2, W // 2) for (r0, nB, H, W) in groups], Ysh


def _sample_offsets(groups):
    off, out = 0, []
    for (_, nB, _, _) in groups:
        out.append(off)
        off += nB
    return out


class BasicLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop=0., attn_drop=0., drop_path=0., norm_layer=nn.LayerNorm, downsample=None):
        super().__init__()
        self.dim, self.input_resolution, self.depth = dim, input_resolution, depth
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, input_resolution, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2, mlp_ratio,
                                 qkv_bias, qk_scale, drop, attn_drop, drop_path[i] if isinstance(drop_path, list) else drop_path,
                                 norm_layer=norm_layer) for i in range(depth)])
        self.downsample = downsample(input_resolution, dim=dim, norm_layer=norm_layer) if downsample is not None else None

    def forward(self, x):
        for blk in self.blocks:
            x, _ = blk(x)
        return self.downsample(x) if self.downsample is not None else x

    def forward_ragged(self, X, groups, shadow=None):
        """X: fp32 token rows [M, C] of several resolution groups, groups: list of (row0, nB, H, W).  The blocks of this
        stage run over all rows at once (Fn.swin_block_multi: attention per group, everything row-wise in one launch),
        then the ragged patch merging.  shadow: the shadow output of the node that produced X (the previous stage's patch merging),
        served by this stage's first block.  Returns (rows of the next stage, its groups, the shadow of those rows or None)."""
        prev_scale = None  # (the producer of X reads dL/dX unscaled)
        nS = sum(g[1] for g in groups)
        # stochastic depth: the per-sample keep factors of every block of this stage (drawn at once by
        # SwinTransformer._draw_drop_path) become per-row scales with ONE gather per stage
        stage_f = []
        for blk in self.blocks:
            pend = None
            if isinstance(blk.drop_path, DropPath) and blk.drop_path.drop_prob and self.training:
                pend = blk.__dict__.pop("_dp_pending", None)
                if pend is None or pend[0].shape[0] != nS:
                    pend = (blk.drop_path.factors(nS, X.device), blk.drop_path.factors(nS, X.device))
            stage_f.append(pend)
        rows_f = None
        live = [i for i, p in enumerate(stage_f) if p is not None]
        if live:
            key = tuple(groups)
            cache = self.__dict__.setdefault("_rowsample", {})
            rowsample = cache.get(key)
            if rowsample is None or rowsample.device != X.device:
                rowsample = torch.cat([torch.arange(nB, device=X.device).repeat_interleave(H * W) + s0
                                       for (_, nB, H, W), s0 in zip(groups, _sample_offsets(groups))])
                cache.clear()
                cache[key] = rowsample
            F = torch.stack([f for i in live for f in stage_f[i]])  # [2 * live blocks, samples]
            rows_f = F[:, rowsample]                                   # [2 * live blocks, M]
        pre = None
        for bi, blk in enumerate(self.blocks):
            dp = None
            if stage_f[bi] is not None:
                k = live.index(bi)
                dp = (rows_f[2 * k], rows_f[2 * k + 1])
            nxt_blk = self.blocks[bi + 1] if bi + 1 < len(self.blocks) else None
            X, shadow, prev_scale, pre = _block_forward_multi(blk, X, groups, dp, shadow, prev_scale, pre, nxt_blk)
        if self.downsample is not None:
            return self.downsample.forward_ragged(X, groups, shadow, prev_scale)
        return X, groups, None

    def forward_with_features(self, x):
        fea = []
        for blk in self.blocks:
            x, _ = blk(x)
            fea.append(x)
        return (self.downsample(x) if self.downsample is not None else x), fea

    def forward_with_attention(self, x):
        attns = []
        for blk in self.blocks:
            x, a = blk(x, return_attention=True)
            attns.append(a)
        return (self.downsample(x) if self.downsample is not None else x), attns


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        img_size, patch_size = _pair(img_size), _pair(patch_size)
        self.img_size, self.patch_size = img_size, patch_size
        self.patches_resolution = [img_size[0] // patch_size[0], img_size[1] // patch_size[1]]
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def forward(self, x):
        g, b = (self.norm.weight, self.norm.bias) if self.norm is not None else (None, None)  # (PATCH_NORM False: no norm)
        return Fn.PatchEmbedFn.apply(x, self.proj.weight, self.proj.bias, g, b, self.patch_size[0])


class SwinTransformer(nn.Module):
    """Same constructor, attributes and methods as the reference class (swin_transformer.py:576-943)."""

    def __init__(self, img_size=224, patch_size=4, in_chans=3, num_classes=1000, embed_dim=96, depths=[2, 2, 6, 2],
                 num_heads=[3, 6, 12, 24], window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0.1, norm_layer=nn.LayerNorm, ape=False, patch_norm=True,
                 use_dense_prediction=False, **kwargs):
        super().__init__()
        self.num_classes, self.num_layers, self.embed_dim = num_classes, len(depths), embed_dim
        self.ape, self.patch_norm, self.mlp_ratio = ape, patch_norm, mlp_ratio
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim, norm_layer if patch_norm else None)
        self.patches_resolution = self.patch_embed.patches_resolution
        if self.ape:  # USE_APE (swin_transformer.py:623-627)
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches, embed_dim))
            _trunc_normal_(self.absolute_pos_embed, std=.02)
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(BasicLayer(
                dim=int(embed_dim * 2 ** i),
                input_resolution=(self.patches_resolution[0] // (2 ** i), self.patches_resolution[1] // (2 ** i)),
                depth=depths[i], num_heads=num_heads[i], window_size=window_size, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate,
                drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])], norm_layer=norm_layer,
                downsample=PatchMerging if i < self.num_layers - 1 else None))
        self.norm = norm_layer(self.num_features)
        self.avgpool = nn.AdaptiveAvgPool1d(1)
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        self.use_dense_prediction = use_dense_prediction
        # run the 224^2 and the 96^2 crops of a step through the backbone together (BasicLayer.forward_multi); False = one pass
        # per resolution group exactly as the reference schedules it (swin_transformer.py:729-751) -- same result either way
        self.ragged_multi_crop = True
        if self.use_dense_prediction:
            self.head_dense = None
        self.apply(self._init_weights)
        # the fused block functions normalise with a compile-time-free but module-independent epsilon (functional.LN_EPS, what
        # get_cls_model passes, swin_transformer.py:963): refuse any other instead of silently computing with the wrong one
        for name, mod in self.named_modules():
            if isinstance(mod, nn.LayerNorm) and abs(mod.eps - Fn.LN_EPS) > 1e-12:
                raise ValueError("LayerNorm %s has eps=%g; the HIP path is built for eps=%g (norm_layer=partial(nn.LayerNorm, eps=1e-6), "
                                 "as models.build_model / get_cls_model construct it)" % (name or "<root>", mod.eps, Fn.LN_EPS))

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            _trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'absolute_pos_embed'}

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {'relative_position_bias_table'}

    # ---- forward paths -----------------------------------------------------------------------
    def _tokens(self, x):
        self._draw_drop_path(x.shape[0], x.device)
        x = self.patch_embed(x)
        if self.ape:
            x = Fn.ApeAddFn.apply(x, self.absolute_pos_embed)
        return x

    def _draw_drop_path(self, nB, device):
        """Stochastic-depth factors floor(keep + U[0,1)) / keep (vision_transformer.py:30-38) for BOTH residual branches
        of EVERY block of this pass in four launches, instead of eight tiny launches per block."""
        blocks = [b for layer in self.layers for b in layer.blocks if isinstance(b.drop_path, DropPath) and b.drop_path.drop_prob]
        if not self.training or not blocks:
            return
        keep = getattr(self, "_dp_keep", None)
        if keep is None or keep.device != device or keep.shape[0] != 2 * len(blocks):
            keep = torch.tensor([1.0 - b.drop_path.drop_prob for b in blocks for _ in range(2)], device=device).unsqueeze(1)
            self.__dict__["_dp_keep"] = keep
        f = torch.rand(2 * len(blocks), nB, device=device).add_(keep).floor_().div_(keep)
        for i, b in enumerate(blocks):
            b.__dict__["_dp_pending"] = (f[2 * i], f[2 * i + 1])

    def forward_feature_maps(self, x):
        x = self._tokens(x)
        for layer in self.layers:
            x = layer(x)
        x_grid = Fn.FinalNormFn.apply(x, self.norm.weight, self.norm.bias)
        return Fn.TokenMeanFn.apply(x_grid), x_grid

    def forward_feature_maps_multi(self, crop_groups):
        """several resolution groups (each a list of equally sized crop batches, in crop order) at once -> (list of
        (cls, region), all region rows [M, C] in group order).  The token rows of all groups travel through the backbone as
        ONE [M, C] matrix (patch embedding, every block, patch merging and the final norm are launched once over all of
        them; only attention, the 2x2 merge gather and the token mean see the grids); the crops are read where they lie
        (no concatenation pass over the images, swin_transformer.py:741)."""
        flat = [c for grp in crop_groups for c in grp]
        self._draw_drop_path(sum(c.shape[0] for c in flat), flat[0].device)
        pe = self.patch_embed
        P = pe.patch_size[0]
        g, b = (pe.norm.weight, pe.norm.bias) if pe.norm is not None else (None, None)
        X = Fn.PatchEmbedMultiFn.apply(pe.proj.weight, pe.proj.bias, g, b, P, *flat)
        groups, r0 = [], 0
        for grp in crop_groups:
            nB, G = sum(c.shape[0] for c in grp), grp[0].shape[-1] // P
            groups.append((r0, nB, G, G))
            r0 += nB * G * G
        shadow = None
        for layer in self.layers:
            X, groups, shadow = layer.forward_ragged(X, groups, shadow)
        C = X.shape[-1]
        Xn = Fn.FinalNormFn.apply(X, self.norm.weight, self.norm.bias)
        parts = torch.split(Xn, [nB * H * W for (_, nB, H, W) in groups]) if len(groups) > 1 else (Xn,)
        outs = []
        for part, (_, nB, H, W) in zip(parts, groups):
            x_grid = part.view(nB, H * W, C)
            outs.append((Fn.TokenMeanFn.apply(x_grid), x_grid))
        return outs, Xn

    def forward_features(self, x):
        cls, region = self.forward_feature_maps(x)
        return (cls, region) if self.use_dense_prediction else cls

    def _apply_head(self, head, feats):
        from ..head import DINOHead
        if isinstance(head, (DINOHead, nn.Identity)):
            return head(feats)
        return head(feats)  # e.g. the default nn.Linear classifier: plain torch module supplied by the caller

    def forward(self, x):
        if not isinstance(x, list):
            x = [x]
        # group consecutive crops of equal resolution (swin_transformer.py:729-732)
        bounds, start = [], 0
        for i in range(1, len(x) + 1):
            if i == len(x) or x[i].shape[-1] != x[start].shape[-1]:
                bounds.append((start, i))
                start = i
        if self.use_dense_prediction:
            cls_parts, fea_parts, npatch = [], [], []
            all_fea = None
            several = len(bounds) > 1 or bounds[0][1] - bounds[0][0] > 1  # (one group of several crops -- the teacher's two global views -- is read where it lies too)
            if self.ragged_multi_crop and several and not self.ape and self._even_maps([x[a].shape[-1] for a, _ in bounds]):
                # every resolution group through the backbone at once (row-wise kernels see all rows, attention runs per group)
                maps, all_fea = self.forward_feature_maps_multi([x[a:b] for a, b in bounds])
            else:
                maps = [self.forward_feature_maps(x[a] if b - a == 1 else torch.cat(x[a:b])) for a, b in bounds]
            for cls, fea in maps:
                B, N, C = fea.shape
                cls_parts.append(cls)
                fea_parts.append(fea.reshape(B * N, C))
                npatch.append(N)
            output_cls = cls_parts[0] if len(cls_parts) == 1 else torch.cat(cls_parts)
            # the ragged route's final-norm output already IS the concatenation of the groups' region rows
            output_fea = all_fea if all_fea is not None else (fea_parts[0] if len(fea_parts) == 1 else torch.cat(fea_parts))
            return self._apply_head(self.head, output_cls), self._apply_head(self.head_dense, output_fea), output_fea, npatch
        outs = [self.forward_features(torch.cat(x[a:b])) for a, b in bounds]
        return self._apply_head(self.head, outs[0] if len(outs) == 1 else torch.cat(outs))

    def _even_maps(self, sizes):
        """the ragged route merges 2 x 2 patches of every group in one launch and has no padding step: crops whose feature map is odd
        at some PatchMerging (e.g. 112^2 at four stages: 28, 14, 7) take the per-group schedule, which pads them as the reference
        does (swin_transformer.py:406-408)"""
        for S in sizes:
            G = S // self.patch_embed.patch_size[0]
            for _ in range(self.num_layers - 1):
                if G % 2:
                    return False
                G //= 2
        return True

    def forward_selfattention(self, x, n=1):
        x = self._tokens(x)
        if n == 1:
            return self.forward_last_selfattention(x)
        return self.forward_all_selfattention(x)

    def forward_last_selfattention(self, x):
        for i, layer in enumerate(self.layers):
            if i < len(self.layers) - 1:
                x = layer(x)
            else:
                x, attns = layer.forward_with_attention(x)
                return attns[-1]

    def forward_all_selfattention(self, x):
        out = []
        for layer in self.layers:
            x, attns = layer.forward_with_attention(x)
            out += attns
        return out

    def forward_return_n_last_blocks(self, x, n=1, return_patch_avgpool=False, depth=[]):
        """mean-pooled tokens of the last n blocks, concatenated (swin_transformer.py:799-837)"""
        start_idx = sum(depth) - n
        acc = 0
        for i, d in enumerate(depth):
            if acc <= start_idx < acc + d:
                start_stage, start_blk = i, start_idx - acc
            acc += d
        x = self._tokens(x)
        output = []
        for i, layer in enumerate(self.layers):
            x, fea = layer.forward_with_features(x)
            if i >= start_stage:
                for x_ in fea[start_blk:]:
                    if i == len(self.layers) - 1:
                        x_ = Fn.FinalNormFn.apply(x_, self.norm.weight, self.norm.bias)
                    output.append(Fn.TokenMeanFn.apply(x_))
                start_blk = 0
        return torch.cat(output, dim=-1)

    def init_weights(self, pretrained='', pretrained_layers=[], verbose=True):
        """Load the keys of a pretrained state_dict this model also has (swin_transformer.py:852-917).  As in the reference
        every such key is loaded (its `need_init` test is always true, whatever `pretrained_layers` says); a
        relative_position_bias_table of another window size and an absolute_pos_embed of another grid are resized
        bicubically (:873-912); any other shape mismatch is an error, as load_state_dict makes it there.  One deliberate
        difference: the derived integer / mask buffers (`relative_position_index`, `attn_mask`) are skipped when their shape
        differs -- the reference means to skip them (:866-867) but its test is always true, so there a checkpoint of another
        window size can never be loaded and the table-resizing branch is unreachable."""
        if not os.path.isfile(pretrained):
            return
        sd = torch.load(pretrained, map_location='cpu')
        logging.info(f'=> loading pretrained model {pretrained}')
        own = self.state_dict()
        picked = {}
        for k, v in sd.items():
            if k not in own:
                continue
            if verbose:
                logging.info(f'=> init {k} from {pretrained}')
            want = own[k].shape
            if ('relative_position_index' in k or 'attn_mask' in k) and v.shape != want:
                continue
            if 'relative_position_bias_table' in k and v.shape != want:
                (L1, nH1), (L2, nH2) = v.shape, want
                if nH1 != nH2:
                    logging.info(f"Error in loading {k}, passing")
                elif L1 != L2:
                    logging.info('=> load_pretrained: resized variant: {} to {}'.format((L1, nH1), (L2, nH2)))
                    S1, S2 = int(L1 ** 0.5), int(L2 ** 0.5)
                    grid = torch.nn.functional.interpolate(v.permute(1, 0).view(1, nH1, S1, S1), size=(S2, S2), mode='bicubic')
                    v = grid.view(nH2, L2).permute(1, 0)
            if 'absolute_pos_embed' in k and v.shape != want:
                (_, L1, C1), (_, L2, C2) = v.shape, want
                if C1 != C2:
                    logging.info(f"Error in loading {k}, passing")
                elif L1 != L2:
                    logging.info('=> load_pretrained: resized variant: {} to {}'.format((1, L1, C1), (1, L2, C2)))
                    S1, S2 = int(L1 ** 0.5), int(L2 ** 0.5)
                    grid = torch.nn.functional.interpolate(v.reshape(-1, S1, S1, C1).permute(0, 3, 1, 2), size=(S2, S2), mode='bicubic')
                    v = grid.permute(0, 2, 3, 1).flatten(1, 2)
            picked[k] = v
        self.load_state_dict(picked, strict=False)

    def freeze_pretrained_layers(self, frozen_layers=[]):
        for name, module in self.named_modules():
            if name.split('.')[0] in frozen_layers or '.'.join(name.split('.')[0:2]) in frozen_layers \
                    or (len(frozen_layers) > 0 and frozen_layers[0] == '*'):
                for p in module.parameters():
                    p.requires_grad = False
        for name, p in self.named_parameters():
            if name.split('.')[0] in frozen_layers or (len(frozen_layers) > 0 and frozen_layers[0] == '*'):
                p.requires_grad = False
        return self


@register_model
def get_cls_model(config, is_teacher=False, use_dense_prediction=False, **kwargs):
    """same config keys as the reference factory (swin_transformer.py:946-980)"""
    spec = config.MODEL.SPEC
    swin = SwinTransformer(
        img_size=config.TRAIN.IMAGE_SIZE[0], in_chans=3, num_classes=config.MODEL.NUM_CLASSES, patch_size=spec['PATCH_SIZE'],
        embed_dim=spec['DIM_EMBED'], depths=spec['DEPTHS'], num_heads=spec['NUM_HEADS'], window_size=spec['WINDOW_SIZE'],
        mlp_ratio=spec['MLP_RATIO'], qkv_bias=spec['QKV_BIAS'], drop_rate=spec['DROP_RATE'],
        attn_drop_rate=spec['ATTN_DROP_RATE'], drop_path_rate=0.0 if is_teacher else spec['DROP_PATH_RATE'],
        norm_layer=partial(nn.LayerNorm, eps=1e-6), ape=spec['USE_APE'], patch_norm=spec['PATCH_NORM'],
        use_dense_prediction=use_dense_prediction)
    if config.MODEL.INIT_WEIGHTS:
        swin.init_weights(config.MODEL.PRETRAINED, config.MODEL.PRETRAINED_LAYERS, config.VERBOSE)
    if config.FINETUNE.FINETUNE:
        swin.freeze_pretrained_layers(config.FINETUNE.FROZEN_LAYERS)
    return swin
