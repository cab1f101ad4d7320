from .swin_transformer import SwinTransformer, get_cls_model  # noqa: F401  (registers 'swin_transformer')
from . import cvt_v4_transformer  # noqa: F401  (registers 'cvt_v4_transformer')
from .registry import is_model, model_entrypoints, register_model  # noqa: F401
from .build import build_model  # noqa: F401
from ..head import DINOHead  # noqa: F401
