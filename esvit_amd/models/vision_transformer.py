"""MI355X-native monolithic ViT backbones behind the reference's module interface (models/vision_transformer.py:96-381:
``VisionTransformer``, ``deit_tiny`` / ``deit_small`` / ``vit_base`` -- main_esvit.py:305-311 builds them by name).

As for Swin, the module tree only HOLDS parameters -- names, shapes, registration order and init are the reference's, so
state_dicts are interchangeable -- and every forward runs esvit_amd.functional: one autograd node per block (LayerNorm, the four
GEMMs with fused epilogues, global attention over the 197 / 37 tokens of a crop on the batched GEMM family + a row softmax, csrc/
vit_attn.hip).  The class token / position embedding (with the reference's bicubic interpolation for the 96^2 crops) are small
torch tensor ops.
"""
import math
from functools import partial

import torch
import torch.nn as nn

from .. import functional as Fn
from ..head import DINOHead  # noqa: F401  (the reference exports it from this module)
from .swin_transformer import DropPath, Mlp, _trunc_normal_


class Attention(nn.Module):
    """vision_transformer.py:67-94 (parameters only)"""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        if qk_scale is not None or attn_drop != 0. or proj_drop != 0.:
            raise NotImplementedError("qk_scale / attention dropout / projection dropout are unused by deit_tiny, deit_small, vit_base")
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)


class Block(nn.Module):
    """vision_transformer.py:96-119"""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0., drop_path=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def _params(self):
        a, m = self.attn, self.mlp
        if a.qkv.bias is None:
            raise NotImplementedError("qkv_bias=False: every reference ViT factory passes qkv_bias=True")
        return [self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias, self.norm2.weight,
                self.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias]

    def _dp(self, nB, device):
        if not isinstance(self.drop_path, DropPath):
            return None
        f1, f2 = self.drop_path.factors(nB, device), self.drop_path.factors(nB, device)  # one draw per residual branch, :114-115
        return None if f1 is None else (f1, f2)

    def forward(self, x, return_attention=False, dp=False):
        """dp: the DropPath factors of the two residual branches, drawn by VisionTransformer for all blocks at once; False = draw here"""
        if return_attention:
            return Fn.vit_block_attention(x, self.attn.num_heads, self._params())
        if dp is False:
            dp = self._dp(x.shape[0], x.device)
        return Fn.vit_block(x, self.attn.num_heads, dp, self._params())

    def forward_fea_and_attn(self, x):
        return self.forward(x), Fn.vit_block_attention(x, self.attn.num_heads, self._params())


class PatchEmbed(nn.Module):
    """vision_transformer.py:121-139"""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.num_patches = (img_size // patch_size) * (img_size // patch_size)
        self.img_size, self.patch_size = img_size, patch_size
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        return Fn.VitPatchEmbedFn.apply(x, self.proj.weight, self.proj.bias, self.patch_size)


class VisionTransformer(nn.Module):
    """vision_transformer.py:142-362"""

    def __init__(self, img_size=[224], patch_size=16, in_chans=3, num_classes=0, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.,
                 qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0., norm_layer=nn.LayerNorm,
                 use_dense_prediction=False, **kwargs):
        super().__init__()
        if drop_rate != 0.:
            raise NotImplementedError("drop_rate: dropout is 0 on the reference's pre-training path")
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(img_size=img_size[0], patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]  # stochastic depth decay rule
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate,
                  attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer) for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        if abs(self.norm.eps - Fn.LN_EPS) > 1e-12:
            raise ValueError("LayerNorm eps %g: the kernels are built for the reference factories' eps = 1e-6" % self.norm.eps)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        self.use_dense_prediction = use_dense_prediction
        if self.use_dense_prediction:
            self.head_dense = None
        _trunc_normal_(self.pos_embed, std=.02)
        _trunc_normal_(self.cls_token, std=.02)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        """vision_transformer.py:177-184: truncated-normal Linear weights, zero biases, unit LayerNorm"""
        if isinstance(m, (nn.Linear, nn.LayerNorm)) and m.bias is not None:
            nn.init.zeros_(m.bias)
        if isinstance(m, nn.Linear):
            _trunc_normal_(m.weight, std=.02)
        elif isinstance(m, nn.LayerNorm):
            nn.init.ones_(m.weight)

    # ---- tokens ------------------------------------------------------------------------------------------------------
    def interpolate_pos_encoding(self, x, pos_embed):
        """vision_transformer.py:263-277.  The patch grid of the position embedding is resampled to the crop's grid with the
        reference's own call -- F.interpolate(scale_factor=sqrt(npatch / N), mode="bicubic") -- so the output size rounds the same
        way; the class-token entry is passed through."""
        n_new, n_old = x.shape[1] - 1, pos_embed.shape[1] - 1
        if n_new == n_old:
            return pos_embed
        side, dim = int(math.sqrt(n_old)), x.shape[-1]
        grid = pos_embed[:, 1:].reshape(1, side, side, dim).permute(0, 3, 1, 2)
        grid = nn.functional.interpolate(grid, scale_factor=math.sqrt(n_new / n_old), mode='bicubic')
        return torch.cat((pos_embed[:, :1], grid.permute(0, 2, 3, 1).reshape(1, -1, dim)), dim=1)

    def _tokens(self, x):
        """patches -> [class token | patch tokens] + position embedding (vision_transformer.py:236-243)"""
        if x.shape[-1] != x.shape[-2]:
            raise NotImplementedError("VisionTransformer: square images only (got %dx%d): the patch embedding and the position-embedding "
                                      "interpolation here assume a square patch grid; the reference's (w0, h0) resampling of "
                                      "vision_transformer.py:236-262 for other shapes is not built" % (x.shape[-2], x.shape[-1]))
        patches = self.patch_embed(x)
        x = torch.cat((self.cls_token.expand(patches.shape[0], -1, -1), patches), dim=1)
        return Fn.ApeAddFn.apply(x, self.interpolate_pos_encoding(x, self.pos_embed).contiguous())

    def _normed(self, x):
        return Fn.FinalNormFn.apply(x, self.norm.weight, self.norm.bias)

    def _drop_path_factors(self, B, device):
        """stochastic depth (vision_transformer.py:30-38) for every block and both residual branches in ONE draw: [depth, 2, B]
        factors floor(keep + u) / keep (four launches per pass instead of 4 x 2 x depth)"""
        if not self.training:
            return None
        rates = [blk.drop_path.drop_prob if isinstance(blk.drop_path, DropPath) else 0.0 for blk in self.blocks]
        if not any(rates):
            return None
        keep = self.__dict__.get("_keep")
        if keep is None or keep.device != device:
            keep = self.__dict__["_keep"] = (1.0 - torch.tensor(rates, dtype=torch.float32, device=device)).view(-1, 1, 1)
        return (keep + torch.rand(len(rates), 2, B, device=device)).floor_().div_(keep)

    def _run_blocks(self, x):
        f = self._drop_path_factors(x.shape[0], x.device)
        for i, blk in enumerate(self.blocks):
            x = blk(x, dp=None if f is None else (f[i, 0], f[i, 1]))
        return x

    def forward_feature_maps(self, x):
        return self._normed(self._run_blocks(self._tokens(x)))

    def forward_features(self, x):
        x = self.forward_feature_maps(x)
        return (x[:, 0], x[:, 1:]) if self.use_dense_prediction else x[:, 0]

    # ---- multi-crop forward (vision_transformer.py:186-233) ---------------------------------------------------------------
    ragged_multi_crop = True  # all resolution groups through one set of LayerNorm / GEMM launches (False: one pass per group)

    def _forward_ragged(self, runs):
        """every run of equal-resolution crops as token ROWS of one matrix: the row-wise kernels of a block see all crops at once,
        the attention is launched per run on its row range (functional.VitBlockMultiFn) -> list of normed [B_g, N_g, C]"""
        toks = [self._tokens(r) for r in runs]
        C = toks[0].shape[-1]
        segs, row0 = [], 0
        for t in toks:
            segs.append((row0, t.shape[0], t.shape[1]))
            row0 += t.shape[0] * t.shape[1]
        X = torch.cat([t.reshape(-1, C) for t in toks])
        nsamp = sum(t.shape[0] for t in toks)
        f = self._drop_path_factors(nsamp, X.device)
        segs = tuple(segs)
        if f is not None:  # rows of a sample share its factor: one gather for all blocks and both branches -> [depth, 2, M]
            cache = self.__dict__.setdefault("_row2sample", {})
            idx = cache.get((segs, X.device))
            if idx is None:  # (built on the host once per batch geometry: no device synchronisation in the step)
                import numpy as np
                idx = cache[(segs, X.device)] = torch.from_numpy(np.repeat(np.arange(nsamp), [n for (_, nB, n) in segs for _ in range(nB)])).to(X.device)
            f = f.index_select(2, idx)
        for i, blk in enumerate(self.blocks):
            X = Fn.vit_block_multi(X, segs, blk.attn.num_heads, None if f is None else (f[i, 0], f[i, 1]), blk._params())
        Xn = self._normed(X.view(1, -1, C)).view(-1, C)
        return [Xn[r0:r0 + nB * n].view(nB, n, C) for (r0, nB, n) in segs]

    def forward(self, x):
        crops = x if isinstance(x, list) else [x]
        runs, start = [], 0
        for i in range(1, len(crops) + 1):
            if i == len(crops) or crops[i].shape[-1] != crops[start].shape[-1]:
                runs.append(torch.cat(crops[start:i]))
                start = i
        if self.ragged_multi_crop and len(runs) > 1:
            maps = self._forward_ragged(runs)
            if not self.use_dense_prediction:
                return self.head(torch.cat([m[:, 0] for m in maps]))
            feats = torch.cat([m[:, 1:].reshape(-1, m.shape[-1]) for m in maps])
            return self.head(torch.cat([m[:, 0] for m in maps])), self.head_dense(feats), feats, [m.shape[1] - 1 for m in maps]
        if not self.use_dense_prediction:
            return self.head(torch.cat([self.forward_features(r) for r in runs]))
        cls, fea, npatch = [], [], []
        for r in runs:
            c, f = self.forward_features(r)
            cls.append(c)
            fea.append(f.reshape(-1, f.shape[-1]))
            npatch.append(f.shape[1])
        feats = torch.cat(fea)
        return self.head(torch.cat(cls)), self.head_dense(feats), feats, npatch

    # ---- evaluation hooks (vision_transformer.py:279-362) ---------------------------------------------------------------
    def forward_selfattention(self, x, n=1):
        """attention probabilities of the last block (n = 1) or of every block; images whose sides are multiples of the patch
        size (the reference's zero-padding branch for other sizes is not needed by its own callers)"""
        x = self._tokens(x)
        maps = []
        for i, blk in enumerate(self.blocks):
            last = i == len(self.blocks) - 1
            if n != 1 or last:
                maps.append(blk(x, return_attention=True))
            if not last:
                x = blk(x)
        return maps[0] if n == 1 else maps

    def forward_return_n_last_blocks(self, x, n=1, return_patch_avgpool=False, depths=[]):
        """eval_linear.py:256,292: class tokens of the n last blocks (each through the final norm), optionally followed by the mean
        patch token of the last block"""
        x = self._tokens(x)
        first = len(self.blocks) - n
        feats = []
        for i, blk in enumerate(self.blocks):
            x = blk(x)
            if i >= first:
                feats.append(self._normed(x))
        out = [f[:, 0] for f in feats]
        if return_patch_avgpool:
            out.append(feats[-1][:, 1:].mean(dim=1))
        return torch.cat(out, dim=-1)


# factories by name (main_esvit.py:305-311 looks them up in this module's __dict__): (embed_dim, num_heads)
_WIDTHS = {"deit_tiny": (192, 3), "deit_small": (384, 6), "vit_base": (768, 12)}


def _factory(name):
    dim, heads = _WIDTHS[name]

    def make(patch_size=16, **kwargs):
        return VisionTransformer(patch_size=patch_size, embed_dim=dim, depth=12, num_heads=heads, mlp_ratio=4, qkv_bias=True,
                                 norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
    make.__name__ = name
    return make


deit_tiny, deit_small, vit_base = _factory("deit_tiny"), _factory("deit_small"), _factory("vit_base")
