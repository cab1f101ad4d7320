"""esvit_amd -- MI355X-native hot path of microsoft/esvit (the EsViT pre-training step).

Public surface mirrors the reference's: ``models.build_model``, ``DINOHead``, ``DINOLoss``/``DDINOLoss``
(same signatures, identical state_dict layout), plus the fused step (``engine.EsvitTrainer``,
``engine.train_one_epoch``) and the fused optimizer (``update.FusedClipAdamWEMA``).  All compute goes through
libesvit_hip.so (hand-written HIP for gfx950); importing this package fails if the library is not built.
"""
import argparse
import os

# ROCm launch-latency knob (read when the HIP runtime initialises, i.e. at the first device call, not at `import torch`):
# kernel arguments are staged in device memory.  The step is ~1500 kernel launches; +0.6 % at B = 128.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch  # noqa: E402

from . import _lib  # noqa: F401  (raises if libesvit_hip.so is missing -- there is no fallback)
from . import models, ops  # noqa: F401
from .head import DINOHead  # noqa: F401
from .loss import DDINOLoss, DINOLoss  # noqa: F401
from .models import build_model  # noqa: F401

# the reference pickles its argparse.Namespace into checkpoints (main_esvit.py:481); torch>=2.6 loads with
# weights_only=True by default and would reject it (SURVEY.md 8b hazard ii)
torch.serialization.add_safe_globals([argparse.Namespace])


def set_precision(name):
    """'bf16' (benchmark mode: bf16 activations/MFMA, fp32 residual stream) or 'fp32' (exact-parity mode)."""
    ops.set_act_dtype({"bf16": torch.bfloat16, "fp32": torch.float32}[name])
    from . import params
    params.clear()
