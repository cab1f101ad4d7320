"""CvT backbone (BASELINE config 5) behind the reference's interface: ``get_cls_model(config, is_teacher,
use_dense_prediction)`` registered as ``cvt_v4_transformer`` (reference: models/cvt_v4_transformer.py:434-700).

The module tree reproduces the reference's parameter / buffer names, shapes and order (``stage{i}.0.proj``,
``stage{i}.1.layers.{j}.0.fn.qkv.{dw,bn,pw}``, ``...1.fn.net.{0,2}``, ``norm``, ``head``), so checkpoints are
interchangeable; the modules are parameter holders only -- the computation runs through the HIP kernels
(``esvit_amd.functional``: ConvEmbedFn, CvtAttnFn, CvtFfnFn) on token-major NHWC activations.

Besides what experiments/imagenet/cvt_v4/s1.yaml uses, the variants of the other cvt_v4 yaml files are built:
``REL_POS_EMBED`` (a relative-position bias table per attention, s1_rpe.yaml) and ``SHIFT`` (s1_shift.yaml: the reference adds
the shifted-window mask of a half-window shift to EVERY block of the stage and never rolls the map -- its ``shift_size`` is
stored and unused, cvt_v4_transformer.py:263,332 -- and that is what runs here) and ``RES_STEM`` (res_stem/*.yaml: three
Conv 3x3 -> BatchNorm -> ReLU units instead of the first ConvEmbed).  Both need every stage's map to be at least
one window wide, as in the reference (its bias / mask shapes do not fit a shrunken window).  Train mode uses batch statistics
(synchronised over the ranks), eval mode the running statistics, as ``nn.BatchNorm2d`` / ``SyncBatchNorm`` do.
"""
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from .. import functional as Fn
from .registry import register_model
from .swin_transformer import DropPath


def _trunc_normal_(t, std=0.02):
    return nn.init.trunc_normal_(t, std=std, a=-2.0, b=2.0)


class LayerNorm(nn.LayerNorm):
    """parameter holder (cvt_v4_transformer.py:35-41)"""


class QuickGELU(nn.Module):
    """x * sigmoid(1.702 x) (cvt_v4_transformer.py:44-46); fused into the fc1 GEMM epilogue"""


class DepthWiseConv2d(nn.Module):
    """dw 3x3 -> BatchNorm2d -> 1x1 (cvt_v4_transformer.py:75-105): parameter holder"""

    def __init__(self, dim_in, dim_out, kernel_size, padding, stride, bias=True):
        super().__init__()
        if kernel_size != 3 or padding != 1 or stride != 1:
            raise NotImplementedError("depthwise qkv convolution: only 3x3 / pad 1 / stride 1 (s1.yaml)")
        self.dw = nn.Conv2d(dim_in, dim_in, kernel_size=kernel_size, padding=padding, groups=dim_in, stride=stride, bias=False)
        self.bn = nn.BatchNorm2d(dim_in)
        self.pw = nn.Conv2d(dim_in, dim_out, kernel_size=1, bias=bias)


class Attention(nn.Module):
    def __init__(self, dim_in, dim_out, num_heads, qkv_bias, kernel_size, padding, window_size, shift_size, rel_pos_embed, **kwargs):
        super().__init__()
        self.heads = num_heads
        self.window_size = window_size
        self.shift_size = shift_size  # (kept like the reference keeps it; nothing reads it)
        self.qkv = DepthWiseConv2d(dim_in, dim_out * 3, kernel_size, padding=padding, stride=1, bias=qkv_bias)
        self.proj_out = nn.Conv2d(dim_out, dim_in, 1)
        self.rel_pos_embed = rel_pos_embed
        if rel_pos_embed:  # cvt_v4_transformer.py:141-163: the Swin index formula, one table row per relative offset
            if window_size * window_size > 64 and dim_out // num_heads != 32:
                raise NotImplementedError("relative-position tables on windows of more than 64 tokens need head_dim 32 (got %d)"
                                          % (dim_out // num_heads))
            c = np.stack(np.meshgrid(np.arange(window_size), np.arange(window_size), indexing="ij")).reshape(2, -1)
            rel = (c[:, :, None] - c[:, None, :]).transpose(1, 2, 0) + (window_size - 1)
            self.register_buffer("rel_pos_idx", torch.from_numpy((rel[:, :, 0] * (2 * window_size - 1) + rel[:, :, 1]).astype(np.int64)))
            self.rel_pos_bias_table = nn.Parameter(torch.zeros((2 * window_size - 1) ** 2, num_heads))
            _trunc_normal_(self.rel_pos_bias_table, std=.02)


class FeedForward(nn.Module):
    def __init__(self, dim, act_layer, mult=4):
        super().__init__()
        self.net = nn.Sequential(nn.Conv2d(dim, int(dim * mult), 1), act_layer(), nn.Conv2d(int(dim * mult), dim, 1))


class PreNorm(nn.Module):
    def __init__(self, norm, dim, fn):
        super().__init__()
        self.norm = norm(dim)
        self.fn = fn


class Transformer(nn.Module):
    def __init__(self, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4., qkv_bias=False, drop_path_rate=None, act_layer=QuickGELU,
                 norm_layer=nn.LayerNorm, kernel_qkv=3, padding_qkv=1, window_size=-1, shift=False, rel_pos_embed=False, **kwargs):
        super().__init__()
        self.layers = nn.ModuleList([])
        for i in range(depth):
            shift_size = window_size // 2 if shift and i % 2 == 1 else 0
            self.layers.append(nn.ModuleList([
                PreNorm(norm_layer, embed_dim,
                        Attention(dim_in=embed_dim, dim_out=embed_dim, num_heads=num_heads, qkv_bias=qkv_bias, kernel_size=kernel_qkv,
                                  padding=padding_qkv, window_size=window_size, shift_size=shift_size, rel_pos_embed=rel_pos_embed)),
                PreNorm(norm_layer, embed_dim, FeedForward(embed_dim, act_layer, mlp_ratio)),
                DropPath(drop_path_rate[i]) if isinstance(drop_path_rate, list) else nn.Identity()]))
        self.window_size = window_size
        self.shift = shift
        self.sync_bn_group = None  # process group of the SyncBatchNorm statistics (None: default group; False: local statistics)

    def drop_path_factors(self, nB, device):
        """stochastic depth (per sample: floor(keep + u) / keep) for every block of the stage and both residual branches in ONE draw:
        [depth, 2, nB] -- four launches per stage and pass instead of eight per block"""
        if not self.training:
            return None
        rates = [dp.drop_prob if isinstance(dp, DropPath) and dp.drop_prob else 0.0 for _, _, dp in self.layers]
        if not any(rates):
            return None
        keep = self.__dict__.get("_keep")
        if keep is None or keep.device != device:
            keep = self.__dict__["_keep"] = (1.0 - torch.tensor(rates, dtype=torch.float32, device=device)).view(-1, 1, 1)
        return (keep + torch.rand(len(rates), 2, nB, device=device)).floor_().div_(keep)

    def forward_tokens(self, x, H, W, feats=None):
        """x fp32 [nB, H*W, C] (token-major); feats: optional list that receives every block's output (forward_with_features)"""
        nB = x.shape[0]
        f = self.drop_path_factors(nB, x.device)
        for li, (attn, ff, drop_path) in enumerate(self.layers):
            dp1 = dp2 = None
            if f is not None and isinstance(drop_path, DropPath) and drop_path.drop_prob:
                dp1, dp2 = f[li, 0], f[li, 1]
            a, bn = attn.fn, attn.fn.qkv.bn
            bn_state = {"group": self.sync_bn_group}
            if self.training:
                if bn.track_running_stats:
                    bn_state.update(running_mean=bn.running_mean, running_var=bn.running_var, num_batches_tracked=bn.num_batches_tracked)
            else:  # nn.BatchNorm2d.eval(): the running statistics
                bn_state.update(eval=True, eval_mean=bn.running_mean, eval_var=bn.running_var)
            rpe = (a.rel_pos_bias_table, a.rel_pos_idx) if a.rel_pos_embed else (None, None)
            x = Fn.CvtAttnFn.apply(x, H, W, a.heads, a.window_size, dp1, bn_state, attn.norm.weight, attn.norm.bias, a.qkv.dw.weight,
                                   bn.weight, bn.bias, a.qkv.pw.weight, a.qkv.pw.bias, a.proj_out.weight, a.proj_out.bias,
                                   rpe[0], rpe[1], bool(self.shift))
            x = Fn.CvtFfnFn.apply(x, dp2, ff.norm.weight, ff.norm.bias, ff.fn.net[0].weight, ff.fn.net[0].bias, ff.fn.net[2].weight,
                                  ff.fn.net[2].bias)
            if feats is not None:
                feats.append(x)
        return x


class ConvEmbed(nn.Module):
    def __init__(self, patch_size=7, in_chans=3, embed_dim=64, stride=4, padding=2, norm_layer=None):
        super().__init__()
        self.patch_size, self.stride, self.padding, self.in_chans = patch_size, stride, padding, in_chans
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=stride, padding=padding)
        self.norm = norm_layer(embed_dim) if norm_layer else None
        if self.norm is None:
            raise NotImplementedError("ConvEmbed without a norm layer is not on the reference path")

    def forward_tokens(self, src, nchw, nB, H, W):
        geo = (nchw, nB, H, W, self.in_chans, self.patch_size, self.stride, self.padding)
        t = Fn.ConvEmbedFn.apply(src, geo, self.proj.weight, self.proj.bias, self.norm.weight, self.norm.bias)
        o = Fn.ops_module()
        return t, o.conv_out_size(H, self.patch_size, self.stride, self.padding), o.conv_out_size(W, self.patch_size, self.stride, self.padding)


class ResStem(nn.Module):
    """two (deep: three) units of Conv2d 3x3 (no bias) -> BatchNorm2d -> ReLU, strides 2 / (1) / 2 (cvt_v4_transformer.py:385-430;
    RES_STEM: True in experiments/imagenet/cvt_v4/res_stem/*.yaml): parameter holder, each unit is one Fn.ConvBnReluFn node"""

    def __init__(self, channels_stem, deep=False):
        super().__init__()
        units, cin = [], 3
        for stride in ((2, 1, 2) if deep else (2, 2)):
            units += [nn.Conv2d(cin, channels_stem, kernel_size=3, stride=stride, padding=1, bias=False), nn.BatchNorm2d(channels_stem),
                      nn.ReLU(inplace=True)]
            cin = channels_stem
        self.stem = nn.Sequential(*units)
        self.sync_bn_group = None  # as Transformer.sync_bn_group

    def forward_tokens(self, src, nchw, nB, H, W):
        assert nchw, "the residual stem reads the images"
        o = Fn.ops_module()
        t = src
        for i in range(0, len(self.stem), 3):
            conv, bn = self.stem[i], self.stem[i + 1]
            st = {"group": self.sync_bn_group}
            if self.training:
                if bn.track_running_stats:
                    st.update(running_mean=bn.running_mean, running_var=bn.running_var, num_batches_tracked=bn.num_batches_tracked)
            else:
                st.update(eval=True, eval_mean=bn.running_mean, eval_var=bn.running_var)
            t = Fn.ConvBnReluFn.apply(t, (i == 0, nB, H, W, conv.in_channels, 3, conv.stride[0], 1), st, conv.weight, bn.weight, bn.bias)
            H, W = o.conv_out_size(H, 3, conv.stride[0], 1), o.conv_out_size(W, 3, conv.stride[0], 1)
        return t.float().view(nB, H * W, -1), H, W


class CvT(nn.Module):
    def __init__(self, *, num_classes, act_layer=QuickGELU, norm_layer=nn.LayerNorm, init='trunc_norm', use_dense_prediction=False, spec=None):
        super().__init__()
        self.num_stages = spec['NUM_STAGES']
        total_depth = sum(spec['DEPTH'])
        dpr = [x.item() for x in torch.linspace(0, spec['DROP_PATH_RATE'], total_depth)]
        in_chans, depth_accum = 3, 0
        for i in range(self.num_stages):
            if i == 0 and getattr(spec, 'RES_STEM', False):  # (getattr like the reference: a plain dict spec never has it)
                conv = ResStem(spec['DIM_EMBED'][i], True)
            else:
                conv = ConvEmbed(patch_size=spec['PATCH_SIZE'][i], in_chans=in_chans, embed_dim=spec['DIM_EMBED'][i],
                                 stride=spec['PATCH_STRIDE'][i], padding=spec['PATCH_PADDING'][i], norm_layer=norm_layer)
            tr = Transformer(embed_dim=spec['DIM_EMBED'][i], depth=spec['DEPTH'][i], num_heads=spec['NUM_HEADS'][i],
                             mlp_ratio=spec['MLP_RATIO'][i], qkv_bias=spec['QKV_BIAS'][i],
                             drop_path_rate=dpr[depth_accum: depth_accum + spec['DEPTH'][i]], act_layer=act_layer, norm_layer=norm_layer,
                             kernel_qkv=spec['KERNEL_QKV'][i], padding_qkv=spec['PADDING_QKV'][i], window_size=spec['WINDOW_SIZE'][i],
                             shift=spec['SHIFT'][i], rel_pos_embed=spec['REL_POS_EMBED'])
            setattr(self, f'stage{i}', nn.Sequential(conv, tr))
            in_chans = spec['DIM_EMBED'][i]
            depth_accum += spec['DEPTH'][i]
        self.norm = norm_layer(in_chans)
        self.num_features = in_chans
        self.head = nn.Linear(in_chans, num_classes) if num_classes > 0 else nn.Identity()
        self.use_dense_prediction = use_dense_prediction
        if self.use_dense_prediction:
            self.head_dense = None
        self.apply(self._init_weights_trunc_normal)

    def _init_weights_trunc_normal(self, m):
        if isinstance(m, (nn.Linear, nn.Conv2d)):
            _trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, (nn.LayerNorm, nn.BatchNorm2d)):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def forward_feature_maps(self, x):
        """images fp32 NCHW -> (cls [nB, C], region tokens [nB, H*W, C]) fp32  (cvt_v4_transformer.py:549-566)"""
        nB, _, H, W = x.shape
        src, nchw = x, True
        for i in range(self.num_stages):
            conv, tr = getattr(self, f'stage{i}')
            t, H, W = conv.forward_tokens(src, nchw, nB, H, W)
            src, nchw = tr.forward_tokens(t, H, W), False
        x_region = Fn.FinalNormFn.apply(src, self.norm.weight, self.norm.bias, Fn.CVT_LN_EPS)
        return Fn.TokenMeanFn.apply(x_region), x_region

    def forward_features(self, x):
        cls, region = self.forward_feature_maps(x)
        return (cls, region) if self.use_dense_prediction else cls

    def forward_return_n_last_blocks(self, x, n=1, return_patch_avgpool=False, depth=[]):
        """token-averaged features of the n last blocks, concatenated (cvt_v4_transformer.py:567-617; eval_linear.py)"""
        start_idx = sum(depth) - n
        sum_cur = 0
        for i, d in enumerate(depth):
            if sum_cur <= start_idx < sum_cur + d:
                start_stage, start_blk = i, start_idx - sum_cur
            sum_cur += d
        nB, _, H, W = x.shape
        src, nchw, output = x, True, []
        for i in range(self.num_stages):
            conv, tr = getattr(self, f'stage{i}')
            t, H, W = conv.forward_tokens(src, nchw, nB, H, W)
            fea = []
            src, nchw = tr.forward_tokens(t, H, W, fea), False
            if i >= start_stage:
                for x_ in fea[start_blk:]:
                    if i == self.num_stages - 1:  # the last stage's features go through the final norm
                        x_ = Fn.FinalNormFn.apply(x_, self.norm.weight, self.norm.bias, Fn.CVT_LN_EPS)
                    output.append(Fn.TokenMeanFn.apply(x_))
                start_blk = 0
        return torch.cat(output, dim=-1)

    def forward(self, x):
        if not isinstance(x, list):
            x = [x]
        bounds, start = [], 0  # consecutive crops of equal resolution run as one batch (cvt_v4_transformer.py:625-628)
        for i in range(1, len(x) + 1):
            if i == len(x) or x[i].shape[-1] != x[start].shape[-1]:
                bounds.append((start, i))
                start = i
        if self.use_dense_prediction:
            cls_parts, fea_parts, npatch = [], [], []
            for a, b in bounds:
                cls, fea = self.forward_feature_maps(torch.cat(x[a:b]))
                B, N, C = fea.shape
                cls_parts.append(cls)
                fea_parts.append(fea.reshape(B * N, C))
                npatch.append(N)
            output_cls = cls_parts[0] if len(cls_parts) == 1 else torch.cat(cls_parts)
            output_fea = fea_parts[0] if len(fea_parts) == 1 else torch.cat(fea_parts)
            return self.head(output_cls), self.head_dense(output_fea), output_fea, npatch
        outs = [self.forward_features(torch.cat(x[a:b])) for a, b in bounds]
        return self.head(outs[0] if len(outs) == 1 else torch.cat(outs))

    def init_weights(self, pretrained='', pretrained_layers=[], verbose=True):
        import os
        if os.path.isfile(pretrained):
            pretrained_dict = torch.load(pretrained, map_location='cpu')
            model_dict = self.state_dict()
            need = {k: v for k, v in pretrained_dict.items() if k in model_dict and (k.split('.')[0] in pretrained_layers or
                                                                                       pretrained_layers[0] == '*')}
            self.load_state_dict(need, strict=False)


@register_model
def get_cls_model(config, is_teacher=False, use_dense_prediction=False, **kwargs):
    cvt_spec = config.MODEL.SPEC
    if is_teacher:
        cvt_spec['DROP_PATH_RATE'] = 0.0
    cvt = CvT(num_classes=config.MODEL.NUM_CLASSES, act_layer=QuickGELU, norm_layer=partial(LayerNorm, eps=1e-5), init='trunc_norm',
              use_dense_prediction=use_dense_prediction, spec=cvt_spec)
    if config.MODEL.INIT_WEIGHTS:
        cvt.init_weights(config.MODEL.PRETRAINED, config.MODEL.PRETRAINED_LAYERS, config.VERBOSE)
    return cvt
