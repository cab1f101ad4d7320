"""TEST INFRASTRUCTURE ONLY -- import the *reference's own* modules from /root/reference.

/root/reference exists only in the build container (never on the GPU box), so this loader is used
(a) by oracle/gen_golden.py to produce the committed fixtures under tests/golden/, and (b) by CPU
tests that are skipped when the reference tree is absent.  Nothing is copied from the reference:
its files are imported where they lie, under the small shims SURVEY.md 8c lists:

  * timm.models.layers {DropPath, to_2tuple, trunc_normal_}  (timm==0.3.2 is not installed)
  * torch._six.container_abcs                                 (removed from modern torch)
  * DINOLoss / DDINOLoss are *executed from the reference source text* of main_esvit.py (class
    bodies located with ``ast``), because importing main_esvit drags in torchvision/timm.data/yacs.
  * a single-process gloo group, since update_center calls dist.all_reduce unguarded.
"""
import ast
import collections.abc
import os
import sys
import types

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

REF_ROOT = os.environ.get("ESVIT_REFERENCE", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REF_ROOT, "models", "swin_transformer.py"))


class _DropPath(nn.Module):
    """per-sample stochastic depth (semantics of vision_transformer.py:30-49)"""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if not self.drop_prob or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = (keep + torch.rand((x.shape[0],) + (1,) * (x.ndim - 1), dtype=x.dtype, device=x.device)).floor_()
        return x.div(keep) * mask


def _install_shims():
    if "timm.models.layers" not in sys.modules:
        timm = types.ModuleType("timm")
        tm = types.ModuleType("timm.models")
        tl = types.ModuleType("timm.models.layers")
        tl.DropPath = _DropPath
        tl.to_2tuple = lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x)
        tl.trunc_normal_ = lambda t, mean=0.0, std=1.0, a=-2.0, b=2.0: nn.init.trunc_normal_(t, mean, std, a, b)
        timm.models, tm.layers = tm, tl
        sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.layers": tl})
    if "torch._six" not in sys.modules:
        six = types.ModuleType("torch._six")
        six.container_abcs = collections.abc
        six.string_classes = (str,)
        sys.modules["torch._six"] = six
        torch._six = six


_CACHE = {}


def load():
    """-> namespace with .models (reference package), .utils, .DINOLoss, .DDINOLoss, .DINOHead"""
    if "ns" in _CACHE:
        return _CACHE["ns"]
    if not available():
        raise RuntimeError("reference tree not found at %s" % REF_ROOT)
    _install_shims()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import models as ref_models          # noqa: the reference's package
        import utils as ref_utils            # noqa
        from models.vision_transformer import DINOHead as RefDINOHead
    src = open(os.path.join(REF_ROOT, "main_esvit.py")).read()
    tree = ast.parse(src)
    env = {"torch": torch, "nn": nn, "F": F, "np": np, "dist": dist}
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name in ("DINOLoss", "DDINOLoss"):
            code = compile(ast.Module(body=[node], type_ignores=[]), os.path.join(REF_ROOT, "main_esvit.py"), "exec")
            exec(code, env)
    ns = types.SimpleNamespace(models=ref_models, utils=ref_utils, DINOHead=RefDINOHead, DINOLoss=env["DINOLoss"],
                               DDINOLoss=env["DDINOLoss"])
    _CACHE["ns"] = ns
    return ns


def load_train_one_epoch():
    """the reference's UNMODIFIED train_one_epoch (main_esvit.py:499-600), compiled from its source text (importing
    main_esvit as a module drags in torchvision / timm.data / yacs), with the globals it uses bound to the reference's utils"""
    import math
    ns = load()
    src = open(os.path.join(REF_ROOT, "main_esvit.py")).read()
    env = {"torch": torch, "nn": nn, "F": F, "np": np, "dist": dist, "math": math, "sys": sys, "os": os, "utils": ns.utils}
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name == "train_one_epoch":
            exec(compile(ast.Module(body=[node], type_ignores=[]), os.path.join(REF_ROOT, "main_esvit.py"), "exec"), env)
    return env["train_one_epoch"]


def load_knn_classifier():
    """the reference's knn_classifier (eval_knn.py:193-232), executed from its source text (importing eval_knn drags in
    torchvision); its hard-wired .cuda() calls are neutralised by the caller (gen_golden.gen_knn)"""
    src = open(os.path.join(REF_ROOT, "eval_knn.py")).read()
    env = {"torch": torch, "nn": nn}
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name == "knn_classifier":
            exec(compile(ast.Module(body=[node], type_ignores=[]), os.path.join(REF_ROOT, "eval_knn.py"), "exec"), env)
    return env["knn_classifier"]


def load_eval_linear():
    """the reference's linear-probe pieces -- train, validate_network, LinearClassifier (eval_linear.py:244-321) -- executed from
    the source text (importing eval_linear drags in torchvision); their .cuda() calls are neutralised by the caller"""
    ns = load()
    src = open(os.path.join(REF_ROOT, "eval_linear.py")).read()
    env = {"torch": torch, "nn": nn, "utils": ns.utils}
    for node in ast.parse(src).body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in ("train", "validate_network", "LinearClassifier"):
            exec(compile(ast.Module(body=[node], type_ignores=[]), os.path.join(REF_ROOT, "eval_linear.py"), "exec"), env)
    return env["train"], env["validate_network"], env["LinearClassifier"]


def ensure_single_process_group():
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("gloo", rank=0, world_size=1)


class AttrDict(dict):
    """yacs-like read-only config: attribute access, AttributeError on missing keys (cvt relies on getattr defaults)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return AttrDict(v) if isinstance(v, dict) else v


def swin_config(embed_dim=96, depths=(2, 2, 6, 2), heads=(3, 6, 12, 24), window=7, drop_path=0.0, img=224, num_classes=0):
    return AttrDict(MODEL=dict(NAME="swin_transformer", NUM_CLASSES=num_classes, INIT_WEIGHTS=False, PRETRAINED="", PRETRAINED_LAYERS=["*"],
                               SPEC=dict(PATCH_SIZE=4, DIM_EMBED=embed_dim, DEPTHS=list(depths), NUM_HEADS=list(heads), WINDOW_SIZE=window,
                                         MLP_RATIO=4, QKV_BIAS=True, DROP_RATE=0, ATTN_DROP_RATE=0, DROP_PATH_RATE=drop_path,
                                         USE_APE=False, PATCH_NORM=True)),
                    TRAIN=dict(IMAGE_SIZE=[img, img]), FINETUNE=dict(FINETUNE=False, FROZEN_LAYERS=[]), VERBOSE=False)


def cvt_config(dims=(64, 192, 384, 768), heads=(1, 3, 6, 12), depths=(2, 2, 6, 2), drop_path=0.0, num_classes=0, rel_pos_embed=False,
               shift=False, res_stem=False, windows=None):
    """experiments/imagenet/cvt_v4/s1.yaml with the stage lists cut to len(dims); rel_pos_embed / shift / res_stem / windows: the
    variants of s1_rpe.yaml, s1_shift.yaml, res_stem/s3_w14.yaml"""
    n = len(dims)
    spec = dict(INIT="trunc_norm", NUM_STAGES=n, REL_POS_EMBED=rel_pos_embed, SHIFT=[shift] * n, DROP_PATH_RATE=drop_path,
                PATCH_SIZE=[7] + [3] * (n - 1), PATCH_STRIDE=[4] + [2] * (n - 1), PATCH_PADDING=[2] + [1] * (n - 1),
                WINDOW_SIZE=list(windows) if windows else [7] * n, DIM_EMBED=list(dims), NUM_HEADS=list(heads), DEPTH=list(depths),
                MLP_RATIO=[4.0] * n, QKV_BIAS=[True] * n, KERNEL_QKV=[3] * n, PADDING_QKV=[1] * n)
    if res_stem:
        spec["RES_STEM"] = True
    return AttrDict(MODEL=dict(NAME="cvt_v4_transformer", NUM_CLASSES=num_classes, INIT_WEIGHTS=False, PRETRAINED="", PRETRAINED_LAYERS=["*"],
                               SPEC=spec), VERBOSE=False)
