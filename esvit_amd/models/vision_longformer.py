"""Vision Longformer (ViL) backbones behind the reference's entry point (models/vision_longformer.py:406-804: ``MsViT`` and
``get_cls_model``, registry name 'vision_longformer'; experiments/imagenet/vil/*/base.yaml).

Four stages.  A stage is a strided patch embedding (Conv2d k = stride = p on the image / on the previous stage's local tokens, a
LayerNorm, the stage's own global tokens in front, a separable absolute position embedding resampled to the crop's grid) followed by
n (AttnBlock, MlpBlock) pairs.  'longformerhand' stages attend through layers/longformer2d.py: every local query sees the global
tokens and the local tokens of its own and the eight adjacent w x w chunks (layers/slidingchunk_2d.py, mode 0, exact 0 -- the
zero-padded and out-of-range positions that implementation masks are exactly the ones that do not exist here), global queries see
everything, and with SHARE_W the global tokens use the same `query` / `kv` / `proj` Linears; s = 0 stages use plain multi-head
attention (vision_longformer.py:36-118).

MI355X formulation: an (AttnBlock, MlpBlock) pair is ONE autograd node on the kernels of the monolithic ViT (functional.VilBlockFn):
LayerNorm, MFMA GEMMs with fused bias / GELU / residual / DropPath epilogues, and for the attention either the fused windowed
kernels (full stages: 197 / 50 / 37 / 9 tokens are one window) or, for the sliding-chunk stages, batched MFMA score GEMMs with a
row softmax restricted to the chunk neighbourhood (esvit_softmax_rows_chunked_fwd).  The patch embeddings are the im2col + GEMM +
LayerNorm node of the CvT path.  Module tree, parameter names and state_dict order are the reference's, so checkpoints interchange.

Built: what the reference's yaml files select -- rpe off (ape = 1), SHARE_W, ATTN_TYPE 'longformerhand' / full, MODE 0, SW_EXACT 0,
no pooled keys, no SE.  Everything else raises NotImplementedError.
"""
import math
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from .. import functional as Fn
from .registry import register_model
from .swin_transformer import DropPath, Mlp, _trunc_normal_


_BICUBIC = {}


# How the position embedding of a stage with global tokens is resampled when the RESAMPLED GRID HAS THE SIZE OF THE ORIGINAL ONE (the
# construction resolution: the reference's scale factor is then sqrt((N + 1) / N) ~ 1.00016, vision_longformer.py:236-262):
#   "device"  what the reference computes where it actually trains and evaluates: torch's CUDA / HIP upsample_bicubic2d returns a copy
#             of its input when output size == input size -> the table is used as it is.  The default: a checkpoint trained by the
#             reference reproduces its features here.
#   "cpu"     what torch's CPU kernel computes (it resamples with the 1.00016 scale: ~1e-3 absolute in a table of std 0.02); the
#             reference fixtures under tests/golden/ were produced on the CPU, so the tests that compare against them select this.
# Real size changes (other crop resolutions) are resampled the same way in both modes.
SAME_SIZE_RESAMPLING = "device"


def _bicubic_matrix(n_in, scale_factor, device):
    """[n_out, n_in] fp32: one axis of torch's upsample_bicubic2d (align_corners False, cubic coefficient -0.75, scales as given:
    source = (dst + 0.5) / scale_factor - 0.5, neighbours clamped to the grid), n_out = floor(n_in * scale_factor)"""
    key = (n_in, float(scale_factor), str(device))
    m = _BICUBIC.get(key)
    if m is None:
        n_out = int(math.floor(n_in * scale_factor))
        A = -0.75
        src = (np.arange(n_out, dtype=np.float64) + 0.5) / scale_factor - 0.5
        x0 = np.floor(src)
        t = src - x0
        w = np.stack([((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A, ((A + 2) * t - (A + 3)) * t * t + 1,
                      ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1, ((A * (2 - t) - 5 * A) * (2 - t) + 8 * A) * (2 - t) - 4 * A], 1)
        M = np.zeros((n_out, n_in), dtype=np.float64)
        for k in range(4):
            np.add.at(M, (np.arange(n_out), np.clip(x0.astype(np.int64) - 1 + k, 0, n_in - 1)), w[:, k])
        m = _BICUBIC[key] = torch.from_numpy(M.astype(np.float32)).to(device)
    return m


class Attention(nn.Module):
    """parameter holder of the full-attention blocks (vision_longformer.py:36-50): qkv, proj"""

    def __init__(self, dim, num_heads=8, qkv_bias=False):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(0.0)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(0.0)
        self.rpe = False


class Long2DSCSelfAttention(nn.Module):
    """parameter holder of the sliding-chunk blocks (layers/longformer2d.py:10-50): query, kv, proj; with shared weights the global
    projections are the SAME modules under a second name, as in the reference (its state_dict lists them twice)"""

    def __init__(self, dim, num_heads=8, qkv_bias=False, w=7, sharew=False, nglo=1):
        super().__init__()
        self.num_heads = num_heads
        self.Nglo = nglo
        self.attention_window = w
        self.query = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        if nglo >= 1:
            if not sharew:
                raise NotImplementedError("Long2DSCSelfAttention: separate global projections (SHARE_W False) are not built")
            self.query_global, self.kv_global, self.proj_global = self.query, self.kv, self.proj
        self.attn_drop = nn.Dropout(0.0)
        self.proj_drop = nn.Dropout(0.0)
        self.rpe = False
        self.mode = 0


class PatchEmbed(nn.Module):
    """vision_longformer.py:188-262"""

    def __init__(self, patch_size, nx, ny, in_chans=3, embed_dim=768, nglo=1, norm_layer=nn.LayerNorm, norm_embed=True, ape=True):
        super().__init__()
        self.patch_size = (patch_size, patch_size)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        if not norm_embed:
            raise NotImplementedError("PatchEmbed without NORM_EMBED: the embedding node normalises (every reference yaml sets it)")
        self.norm_embed = norm_layer(embed_dim)
        self.nx, self.ny, self.Nglo = nx, ny, nglo
        if nglo >= 1:
            self.cls_token = nn.Parameter(torch.zeros(1, nglo, embed_dim))
            _trunc_normal_(self.cls_token, std=.02)
        else:
            self.cls_token = None
        self.ape = ape
        if not ape:
            raise NotImplementedError("ViL stages with a0 (relative position bias tables instead of the absolute embedding) are not built")
        self.cls_pos_embed = nn.Parameter(torch.zeros(1, nglo, embed_dim))
        self.x_pos_embed = nn.Parameter(torch.zeros(1, nx, embed_dim // 2))
        self.y_pos_embed = nn.Parameter(torch.zeros(1, ny, embed_dim // 2))
        for p in (self.cls_pos_embed, self.x_pos_embed, self.y_pos_embed):
            _trunc_normal_(p, std=.02)
        self.pos_drop = nn.Dropout(p=0.0)

    def position_embedding(self, ntok):
        """[1, Nglo + nx' ny', C] for a crop whose token count (global tokens included) is ntok (vision_longformer.py:236-262: the
        reference compares the count WITH the global tokens against nx ny, so the grid is resampled -- with its own F.interpolate
        call, scale factor sqrt(ntok / (nx ny)) -- even at the construction resolution when a global token exists)"""
        pos = torch.cat([self.x_pos_embed.unsqueeze(2).expand(-1, -1, self.ny, -1),
                         self.y_pos_embed.unsqueeze(1).expand(-1, self.nx, -1, -1)], dim=-1).flatten(start_dim=1, end_dim=2)
        N = pos.shape[1]
        if ntok != N:
            dim, side = pos.shape[-1], int(math.sqrt(N))
            # F.interpolate(scale_factor = sqrt(ntok / N), mode='bicubic') as torch's CPU kernel evaluates it (the reference fixtures):
            # written out as two small matrix products, because torch's device kernel copies its input whenever the output SIZE
            # equals the input size -- which is the case at the construction resolution, where the reference's scale factor is
            # sqrt((N + 1) / N) ~ 1.00016 and its CPU path does resample (1e-3 absolute in the embedding)
            sf = math.sqrt(ntok / N)
            if SAME_SIZE_RESAMPLING == "cpu" or int(math.floor(side * sf)) != side:
                R = _bicubic_matrix(side, sf, pos.device)
                pos = torch.einsum('oh,bhwc,pw->bopc', R, pos.reshape(1, side, side, dim), R).contiguous().view(1, -1, dim)
        return torch.cat([self.cls_pos_embed, pos], dim=1)

    def forward(self, src, nB, H, W, nchw):
        """src: fp32 NCHW images (first stage) or fp32 tokens [nB, H W, Cin] of the previous stage -> ([nB, Nglo + nx ny, C], nx, ny)"""
        k = self.patch_size[0]
        Cin = self.proj.weight.shape[1]
        geo = (bool(nchw), nB, H, W, Cin, k, k, 0, self.norm_embed.eps)
        x = Fn.ConvEmbedFn.apply(src, geo, self.proj.weight, self.proj.bias, self.norm_embed.weight, self.norm_embed.bias)
        nx, ny = H // k, W // k
        if self.cls_token is not None:
            x = torch.cat((self.cls_token.expand(nB, -1, -1), x), dim=1)
        pos = self.position_embedding(x.shape[1])
        if pos.shape[1] != x.shape[1]:
            raise ValueError("position embedding of %d tokens does not fit a crop of %d (the reference's broadcast add fails the same way)"
                             % (pos.shape[1], x.shape[1]))
        return Fn.ApeAddFn.apply(x, pos.contiguous()), nx, ny


class AttnBlock(nn.Module):
    """vision_longformer.py:283-375 (parameter holder: norm, attn, drop_path)"""

    def __init__(self, dim, num_heads, qkv_bias=True, drop_path=0., norm_layer=nn.LayerNorm, attn_type='full', w=7, sharew=False, nglo=1):
        super().__init__()
        self.norm = norm_layer(dim)
        if attn_type == 'full':
            self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        elif attn_type == 'longformerhand':
            self.attn = Long2DSCSelfAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias, w=w, sharew=sharew, nglo=nglo)
        else:
            raise NotImplementedError("ViL attention type %r: 'longformerhand' and full attention are built" % (attn_type,))
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.se = None


class MlpBlock(nn.Module):
    """vision_longformer.py:378-403"""

    def __init__(self, dim, mlp_ratio=4., drop_path=0., norm_layer=nn.LayerNorm):
        super().__init__()
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio))
        self.shortcut = nn.Identity()


def _pair_params(ab, mb):
    a = ab.attn
    if isinstance(a, Long2DSCSelfAttention):
        proj = (a.query.weight, a.query.bias, a.kv.weight, a.kv.bias)
    else:
        proj = (a.qkv.weight, a.qkv.bias, None, None)
    return (ab.norm.weight, ab.norm.bias) + proj + (a.proj.weight, a.proj.bias, mb.norm.weight, mb.norm.bias,
                                                      mb.mlp.fc1.weight, mb.mlp.fc1.bias, mb.mlp.fc2.weight, mb.mlp.fc2.bias)


class MsViT(nn.Module):
    """Multiscale Vision Transformer (vision_longformer.py:406-770): same constructor surface for what is built"""

    def __init__(self, arch, img_size=512, in_chans=3, num_classes=1000, qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., norm_layer=partial(nn.LayerNorm, eps=1e-6), norm_embed=False, w=7, d=1, sharew=False, only_glo=False,
                 share_kv=False, attn_type='longformerhand', sw_exact=0, mode=0, pool_method=None, use_dense_prediction=False,
                 with_se=False, se_mlp_ratio=0.625, se_mlp_balance=False, **args):
        super().__init__()
        if drop_rate or attn_drop_rate or qk_scale or only_glo or sw_exact or mode or pool_method or with_se:
            raise NotImplementedError("MsViT: dropout, qk_scale, ONLY_GLOBAL, SW_EXACT != 0, MODE != 0, pooled keys and SE layers are not built "
                                      "(no reference yaml selects them)")
        self.num_classes = num_classes
        if 'ln_eps' in args:
            norm_layer = partial(nn.LayerNorm, eps=args['ln_eps'])
        self.norm_layer = norm_layer
        if abs(norm_layer(4).eps - Fn.LN_EPS) > 1e-12:
            raise ValueError("LayerNorm eps %g: the block kernels are built for the reference yamls' LN_EPS = 1e-6" % norm_layer(4).eps)
        self.drop_path_rate = drop_path_rate
        self.attn_type = attn_type
        self.sharew = sharew
        self.norm_embed = norm_embed
        self.qkv_bias = qkv_bias
        self.Nx = self.Ny = img_size

        def parse_arch(arch):
            cfgs = []
            for layer in arch.split('_'):
                cfg = {'l': 1, 'h': 3, 'd': 192, 'n': 1, 's': 1, 'g': 1, 'p': 2, 'f': 7, 'a': 1, 'r': 0}  # defaults
                for attr in layer.split(','):
                    cfg[attr[0]] = int(attr[1:])
                cfgs.append(cfg)
            return cfgs

        self.layer_cfgs = parse_arch(arch)
        self.num_layers = len(self.layer_cfgs)
        self.depth = sum(cfg['n'] for cfg in self.layer_cfgs)
        self.out_planes = self.layer_cfgs[-1]['d']
        self.num_features = self.out_planes
        self.Nglos = [cfg['g'] for cfg in self.layer_cfgs]
        self.avg_pool = args['avg_pool'] if 'avg_pool' in args else False
        dprs = torch.linspace(0, drop_path_rate, self.depth).split([cfg['n'] for cfg in self.layer_cfgs])
        if self.num_layers not in (3, 4):
            raise ValueError("Numer of layers {} not implemented yet!".format(self.num_layers))
        self.layer1 = self._make_layer(in_chans, self.layer_cfgs[0], dprs[0], 1)
        self.layer2 = self._make_layer(self.layer_cfgs[0]['d'], self.layer_cfgs[1], dprs[1], 2)
        self.layer3 = self._make_layer(self.layer_cfgs[1]['d'], self.layer_cfgs[2], dprs[2], 3)
        self.layer4 = self._make_layer(self.layer_cfgs[2]['d'], self.layer_cfgs[3], dprs[3], 4) if self.num_layers == 4 else None
        self.norm = norm_layer(self.out_planes)
        self.head = nn.Linear(self.out_planes, num_classes) if num_classes > 0 else nn.Identity()
        self.use_dense_prediction = use_dense_prediction
        if self.use_dense_prediction:
            self.head_dense = None
        self.apply(self._init_weights)

    def _make_layer(self, in_dim, cfg, dprs, layerid):
        assert layerid == cfg['l'], "Error in _make_layer: layerid {} does not equal to layer_id {}".format(layerid, cfg['l'])
        if cfg['r']:
            raise NotImplementedError("ViL stages with pooled keys (r1) are not built")
        self.Nx = nx = self.Nx // cfg['p']
        self.Ny = ny = self.Ny // cfg['p']
        attn_type = self.attn_type if cfg['s'] else 'full'
        layers = [PatchEmbed(cfg['p'], nx, ny, in_chans=in_dim, embed_dim=cfg['d'], nglo=cfg['g'], norm_layer=self.norm_layer,
                             norm_embed=self.norm_embed, ape=bool(cfg['a']))]
        for dpr in dprs:
            layers.append(AttnBlock(cfg['d'], cfg['h'], qkv_bias=self.qkv_bias, drop_path=float(dpr), norm_layer=self.norm_layer,
                                    attn_type=attn_type, w=cfg['f'], sharew=self.sharew, nglo=cfg['g']))
            layers.append(MlpBlock(cfg['d'], drop_path=float(dpr), norm_layer=self.norm_layer))
        return nn.Sequential(*layers)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            _trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def no_weight_decay(self):
        return {'pos_embed', 'cls_token', 'norm.weight', 'norm.bias', 'norm_embed', 'head.bias', 'relative_position'}

    def get_classifier(self):
        return self.head

    # ---- stages -------------------------------------------------------------------------------------------------------
    _CHUNKS = {}

    @classmethod
    def _chunk_table(cls, nglo, nx, ny, w, device):
        """token -> -1 (global) | (chunk row << 16 | chunk column), tokens in the reference's order [globals | (x y) row-major]"""
        key = (nglo, nx, ny, w, str(device))
        t = cls._CHUNKS.get(key)
        if t is None:
            ix, iy = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
            loc = ((ix // w) << 16 | (iy // w)).reshape(-1)
            t = cls._CHUNKS[key] = torch.from_numpy(np.concatenate([np.full(nglo, -1), loc]).astype(np.int32)).to(device)
        return t

    def _stage(self, layer, cfg, src, nB, H, W, nchw, collect=None):
        x, nx, ny = layer[0](src, nB, H, W, nchw)
        sparse = isinstance(layer[1].attn, Long2DSCSelfAttention)
        # (chunk table, global tokens, tokens per chunk row): the local tokens are ordered (x, y), i.e. chunk row by chunk row
        chunk = (self._chunk_table(cfg['g'], nx, ny, cfg['f'], x.device), cfg['g'], cfg['f'] * ny) if sparse else None
        for b in range(1, len(layer), 2):
            ab, mb = layer[b], layer[b + 1]
            dp = None
            if self.training and isinstance(ab.drop_path, DropPath) and ab.drop_path.drop_prob > 0:
                keep = 1.0 - ab.drop_path.drop_prob
                f = (keep + torch.rand(2, nB, device=x.device)).floor_().div_(keep)  # (vision_transformer.py:30-38, per sample)
                dp = (f[0], f[1])
            x = Fn.vil_block(x, cfg['h'], dp, chunk, _pair_params(ab, mb))
            if collect is not None:
                collect.append(x)
        return x, nx, ny

    def _layers(self):
        return [l for l in (self.layer1, self.layer2, self.layer3, self.layer4) if l is not None]

    def forward_feature_maps(self, img):
        """-> LayerNorm-ed tokens of the last stage [nB, Nglo + nx ny, C]"""
        nB, _, H, W = img.shape
        src, nchw = img, True
        for i, (layer, cfg) in enumerate(zip(self._layers(), self.layer_cfgs)):
            x, nx, ny = self._stage(layer, cfg, src, nB, H, W, nchw)
            if i + 1 < self.num_layers:  # the next embedding convolves the local tokens (vision_longformer.py:583-590)
                src, H, W, nchw = x[:, cfg['g']:].contiguous(), nx, ny, False
        return Fn.FinalNormFn.apply(x, self.norm.weight, self.norm.bias)

    def forward_features(self, img):
        x = self.forward_feature_maps(img)
        if self.Nglos[-1] > 0 and not self.avg_pool:
            x_cls, x_region = x[:, 0], x[:, 1:]
        else:
            x_cls, x_region = Fn.TokenMeanFn.apply(x), x
        return (x_cls, x_region) if self.use_dense_prediction else x_cls

    def forward_return_n_last_blocks(self, x, n=1, return_patch_avgpool=False, depth=[]):
        """vision_longformer.py:617-660: the class token (or the token mean of a stage without one) after each of the last n blocks,
        the last stage through the final norm; `depth` = blocks per stage"""
        num_blks = sum(depth)
        start_idx = num_blks - n
        sum_cur = 0
        for i, d in enumerate(depth):
            if sum_cur <= start_idx < sum_cur + d:
                start_stage, start_blk = i, start_idx - sum_cur
            sum_cur += d
        output = []
        nB, _, H, W = x.shape
        src, nchw = x, True
        for i, (layer, cfg) in enumerate(zip(self._layers(), self.layer_cfgs)):
            fea = []
            t, nx, ny = self._stage(layer, cfg, src, nB, H, W, nchw, collect=fea)
            if i + 1 < self.num_layers:
                src, H, W, nchw = t[:, cfg['g']:].contiguous(), nx, ny, False
            if i >= start_stage:
                for x_ in fea[start_blk:]:
                    if i == self.num_layers - 1:
                        x_ = Fn.FinalNormFn.apply(x_, self.norm.weight, self.norm.bias)
                    output.append(x_[:, 0] if (self.Nglos[i] > 0 and not self.avg_pool) else Fn.TokenMeanFn.apply(x_))
                start_blk = 0
        return torch.cat(output, dim=-1)

    # ---- ragged multi-crop route: the full-attention stages see the token rows of all resolution groups at once ------------------
    ragged_multi_crop = True  # (False: one pass per resolution group, the reference's schedule)

    def _forward_ragged(self, runs):
        """-> LayerNorm-ed last-stage tokens per group.  Patch embeddings and sliding-chunk stages run per group (their geometry
        differs); in the full-attention stages LayerNorm, the four GEMMs and the residual adds of a block pair are row-wise, so the rows
        of all groups go through ONE set of launches (functional.vit_block_multi: the attention per group on its row range) -- half the
        launches there, one gradient contribution per parameter, half the split-K partial traffic (cf. the ViT route, section 10)"""
        dev = runs[0].device
        st = [dict(src=r, nB=r.shape[0], H=r.shape[2], W=r.shape[3], nchw=True) for r in runs]
        nsamp = sum(g["nB"] for g in st)
        outs = None
        for i, (layer, cfg) in enumerate(zip(self._layers(), self.layer_cfgs)):
            sparse = isinstance(layer[1].attn, Long2DSCSelfAttention)
            if sparse:
                xs = []
                for g in st:
                    x, nx, ny = self._stage(layer, cfg, g["src"], g["nB"], g["H"], g["W"], g["nchw"])
                    xs.append(x)
                    g["nx"], g["ny"] = nx, ny
            else:
                toks = []
                for g in st:
                    t, nx, ny = layer[0](g["src"], g["nB"], g["H"], g["W"], g["nchw"])
                    toks.append(t)
                    g["nx"], g["ny"] = nx, ny
                C = toks[0].shape[-1]
                segs, row0 = [], 0
                for t in toks:
                    segs.append((row0, t.shape[0], t.shape[1]))
                    row0 += t.shape[0] * t.shape[1]
                segs = tuple(segs)
                X = torch.cat([t.reshape(-1, C) for t in toks])
                idx = None
                for b in range(1, len(layer), 2):
                    ab, mb = layer[b], layer[b + 1]
                    dp = None
                    if self.training and isinstance(ab.drop_path, DropPath) and ab.drop_path.drop_prob > 0:
                        keep = 1.0 - ab.drop_path.drop_prob
                        f = (keep + torch.rand(2, nsamp, device=dev)).floor_().div_(keep)
                        if idx is None:  # rows of a sample share its factor (built on the host once per batch geometry)
                            cache = self.__dict__.setdefault("_row2sample", {})
                            idx = cache.get((segs, str(dev)))
                            if idx is None:
                                idx = cache[(segs, str(dev))] = torch.from_numpy(np.repeat(np.arange(nsamp), [n for (_, nB, n) in segs for _ in range(nB)])).to(dev)
                        fr = f.index_select(1, idx)
                        dp = (fr[0], fr[1])
                    a = ab.attn
                    prm = (ab.norm.weight, ab.norm.bias, a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias, mb.norm.weight, mb.norm.bias,
                           mb.mlp.fc1.weight, mb.mlp.fc1.bias, mb.mlp.fc2.weight, mb.mlp.fc2.bias)
                    X = Fn.vit_block_multi(X, segs, cfg['h'], dp, prm)
                if i + 1 == self.num_layers:
                    X = Fn.FinalNormFn.apply(X.view(1, -1, C), self.norm.weight, self.norm.bias).view(-1, C)
                xs = [X[r0:r0 + nB * n].view(nB, n, C) for (r0, nB, n) in segs]
            if i + 1 < self.num_layers:
                for g, x in zip(st, xs):
                    g.update(src=x[:, cfg['g']:].contiguous(), H=g["nx"], W=g["ny"], nchw=False)
            else:
                outs = xs if not sparse else [Fn.FinalNormFn.apply(x, self.norm.weight, self.norm.bias) for x in xs]
        return outs

    def _split_features(self, x):
        if self.Nglos[-1] > 0 and not self.avg_pool:
            return x[:, 0], x[:, 1:]
        return Fn.TokenMeanFn.apply(x), x

    # ---- multi-crop forward (vision_longformer.py:699-752) ---------------------------------------------------------------
    def forward(self, x):
        crops = x if isinstance(x, list) else [x]
        runs, start = [], 0
        for i in range(1, len(crops) + 1):
            if i == len(crops) or crops[i].shape[-1] != crops[start].shape[-1]:
                runs.append(torch.cat(crops[start:i]))
                start = i
        if self.ragged_multi_crop and len(runs) > 1:
            parts = [self._split_features(m) for m in self._forward_ragged(runs)]
            if not self.use_dense_prediction:
                return self.head(torch.cat([c for c, _ in parts]))
            feats = torch.cat([f.reshape(-1, f.shape[-1]) for _, f in parts])
            return self.head(torch.cat([c for c, _ in parts])), self.head_dense(feats), feats, [f.shape[1] for _, f in parts]
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


VIL_SPECS = {
    # experiments/imagenet/vil/vil_tiny/base.yaml, vil_small/base.yaml
    "vil_tiny": 'l1,h1,d48,n1,s1,g1,p4,f7_l2,h3,d96,n1,s1,g1,p2,f7_l3,h3,d192,n9,s0,g1,p2,f7_l4,h6,d384,n1,s0,g0,p2,f7',
    "vil_small": 'l1,h3,d96,n1,s1,g1,p4,f7_l2,h3,d192,n2,s1,g1,p2,f7_l3,h6,d384,n8,s0,g1,p2,f7_l4,h12,d768,n1,s0,g0,p2,f7',
}


@register_model
def get_cls_model(config, is_teacher=False, use_dense_prediction=False, **kwargs):
    """vision_longformer.py:755-790"""
    spec = config.MODEL.SPEC
    ms = spec.MSVIT

    def opt(node, key, default):
        v = getattr(node, key, default)
        return default if v in (None, "None") else v

    return MsViT(arch=ms.ARCH, img_size=config.TRAIN.IMAGE_SIZE[0], num_classes=config.MODEL.NUM_CLASSES,
                 drop_rate=opt(spec, "DROP", 0.0), drop_path_rate=0.0 if is_teacher else opt(spec, "DROP_PATH", 0.0),
                 norm_embed=opt(spec, "NORM_EMBED", False), avg_pool=opt(spec, "AVG_POOL", False),
                 sharew=opt(ms, "SHARE_W", False), attn_type=opt(ms, "ATTN_TYPE", "longformerhand"), share_kv=opt(ms, "SHARE_KV", False),
                 only_glo=opt(ms, "ONLY_GLOBAL", False), sw_exact=opt(ms, "SW_EXACT", 0), ln_eps=opt(ms, "LN_EPS", 1e-6),
                 mode=opt(ms, "MODE", 0), pool_method=opt(ms, "POOL_METHOD", None), with_se=opt(ms, "WITH_SE", False),
                 use_dense_prediction=use_dense_prediction)
