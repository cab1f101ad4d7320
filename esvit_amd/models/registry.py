"""Model registry keyed by the defining module's file name (reference: models/registry.py:4-18), so that
``config.MODEL.NAME == 'swin_transformer'`` resolves exactly as in the reference."""
_ENTRYPOINTS = {}


def register_model(fn):
    _ENTRYPOINTS[fn.__module__.split(".")[-1]] = fn
    return fn


_ALIASES = {"cls_vil": "vision_longformer"}  # experiments/imagenet/vil/vil_tiny/base.yaml names the model this way


def model_entrypoints(model_name):
    return _ENTRYPOINTS[_ALIASES.get(model_name, model_name)]


def is_model(model_name):
    return _ALIASES.get(model_name, model_name) in _ENTRYPOINTS
