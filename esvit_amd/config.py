"""Minimal config object with the attribute surface the model factory reads (config/default.py keys
MODEL.NAME / MODEL.SPEC.* / MODEL.NUM_CLASSES / MODEL.INIT_WEIGHTS / TRAIN.IMAGE_SIZE / FINETUNE.* / VERBOSE),
buildable from a reference yaml (experiments/imagenet/swin/*.yaml) without yacs."""
import yaml


class CfgNode(dict):
    """attribute access; missing keys raise AttributeError (so ``getattr(spec, 'X', default)`` works as with yacs)"""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __setattr__(self, k, v):
        self[k] = v


_DEFAULTS = dict(
    MODEL=dict(NAME="swin_transformer", NUM_CLASSES=0, INIT_WEIGHTS=False, PRETRAINED="", PRETRAINED_LAYERS=["*"], SPEC={}),
    TRAIN=dict(IMAGE_SIZE=[224, 224]), FINETUNE=dict(FINETUNE=False, FROZEN_LAYERS=[]), VERBOSE=False)

SWIN_SPECS = {
    "swin_tiny_w7": dict(PATCH_SIZE=4, DIM_EMBED=96, DEPTHS=[2, 2, 6, 2], NUM_HEADS=[3, 6, 12, 24], WINDOW_SIZE=7, MLP_RATIO=4,
                         QKV_BIAS=True, DROP_RATE=0, ATTN_DROP_RATE=0, DROP_PATH_RATE=0.1, USE_APE=False, PATCH_NORM=True),
    "swin_tiny_w14": dict(PATCH_SIZE=4, DIM_EMBED=96, DEPTHS=[2, 2, 6, 2], NUM_HEADS=[3, 6, 12, 24], WINDOW_SIZE=14, MLP_RATIO=4,
                          QKV_BIAS=True, DROP_RATE=0, ATTN_DROP_RATE=0, DROP_PATH_RATE=0.1, USE_APE=False, PATCH_NORM=True),
    "swin_base_w14": dict(PATCH_SIZE=4, DIM_EMBED=128, DEPTHS=[2, 2, 18, 2], NUM_HEADS=[4, 8, 16, 32], WINDOW_SIZE=14, MLP_RATIO=4,
                          QKV_BIAS=True, DROP_RATE=0, ATTN_DROP_RATE=0, DROP_PATH_RATE=0.2, USE_APE=False, PATCH_NORM=True),
    "swin_small_w7": dict(PATCH_SIZE=4, DIM_EMBED=96, DEPTHS=[2, 2, 18, 2], NUM_HEADS=[3, 6, 12, 24], WINDOW_SIZE=7, MLP_RATIO=4,
                          QKV_BIAS=True, DROP_RATE=0, ATTN_DROP_RATE=0, DROP_PATH_RATE=0.2, USE_APE=False, PATCH_NORM=True),
    "swin_base_w7": dict(PATCH_SIZE=4, DIM_EMBED=128, DEPTHS=[2, 2, 18, 2], NUM_HEADS=[4, 8, 16, 32], WINDOW_SIZE=7, MLP_RATIO=4,
                         QKV_BIAS=True, DROP_RATE=0, ATTN_DROP_RATE=0, DROP_PATH_RATE=0.2, USE_APE=False, PATCH_NORM=True),
}


# experiments/imagenet/cvt_v4/s1.yaml (CvT-13: BASELINE config 5)
CVT_SPECS = {
    "cvt_s1": dict(INIT="trunc_norm", NUM_STAGES=4, REL_POS_EMBED=False, SHIFT=[False] * 4, DROP_PATH_RATE=0.1, PATCH_SIZE=[7, 3, 3, 3],
                   PATCH_STRIDE=[4, 2, 2, 2], PATCH_PADDING=[2, 1, 1, 1], WINDOW_SIZE=[7, 7, 7, 7], DIM_EMBED=[64, 192, 384, 768],
                   NUM_HEADS=[1, 3, 6, 12], DEPTH=[2, 2, 6, 2], MLP_RATIO=[4.0] * 4, QKV_BIAS=[True] * 4, KERNEL_QKV=[3] * 4,
                   PADDING_QKV=[1] * 4),
}


def _merge(base, over):
    out = dict(base)
    for k, v in over.items():
        out[k] = _merge(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else v
    return out


def swin_config(name="swin_tiny_w7", **spec_overrides):
    spec = dict(SWIN_SPECS[name])
    spec.update(spec_overrides)
    cfg = _merge(_DEFAULTS, dict(MODEL=dict(SPEC=spec)))
    return CfgNode(cfg)


def cvt_config(name="cvt_s1", **spec_overrides):
    spec = dict(CVT_SPECS[name])
    spec.update(spec_overrides)
    cfg = _merge(_DEFAULTS, dict(MODEL=dict(NAME="cvt_v4_transformer", SPEC=spec)))
    return CfgNode(cfg)


# experiments/imagenet/vil/{vil_tiny,vil_small}/base.yaml (Vision Longformer; ARCH strings: esvit_amd/models/vision_longformer.py)
VIL_MSVIT = dict(LN_EPS=1e-6, SHARE_W=True, ATTN_TYPE="longformerhand", SHARE_KV=True, ONLY_GLOBAL=False, SW_EXACT=0, MODE=0,
                 VIL_MODE_SWITCH=0.5, POOL_METHOD=None, WITH_SE=None)


def vil_config(name="vil_tiny", arch=None, **spec_overrides):
    from .models.vision_longformer import VIL_SPECS
    spec = dict(AVG_POOL=False, DROP=0.0, DROP_PATH=0.1, NORM_EMBED=True, MSVIT=dict(VIL_MSVIT, ARCH=arch or VIL_SPECS[name]))
    spec.update(spec_overrides)
    cfg = _merge(_DEFAULTS, dict(MODEL=dict(NAME="vision_longformer", SPEC=spec)))
    return CfgNode(cfg)


def model_config(name, **spec_overrides):
    """named architecture -> config (swin_* -> swin_transformer, cvt_* -> cvt_v4_transformer, vil_* -> vision_longformer)"""
    if name.startswith("vil_"):
        return vil_config(name, **spec_overrides)
    return cvt_config(name, **spec_overrides) if name in CVT_SPECS else swin_config(name, **spec_overrides)


def from_yaml(path, opts=None):
    """read a reference experiment yaml (no BASE inheritance needed for swin yamls); opts = [KEY, VALUE, ...]"""
    with open(path) as f:
        cfg = _merge(_DEFAULTS, yaml.safe_load(f) or {})
    for k, v in zip((opts or [])[0::2], (opts or [])[1::2]):
        node = cfg
        parts = k.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = yaml.safe_load(v) if isinstance(v, str) else v
    return CfgNode(cfg)
