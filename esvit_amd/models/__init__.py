from .swin_transformer import SwinTransformer, get_cls_model  # noqa: F401  (registers 'swin_transformer')
from . import cvt_v4_transformer  # noqa: F401  (registers 'cvt_v4_transformer')
from . import vision_longformer  # noqa: F401  (registers 'vision_longformer': MsViT)
from . import vision_transformer  # noqa: F401  (deit_tiny / deit_small / vit_base, built by name: main_esvit.py:305-311)
from .registry import is_model, model_entrypoints, register_model  # noqa: F401
from .build import build_model  # noqa: F401
from ..head import DINOHead  # noqa: F401
