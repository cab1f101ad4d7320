"""Backbone factory behind the reference's entry point ``models.build_model`` (models/build.py:5-10).

The reference resolves ``config.MODEL.NAME`` through a registry of per-file constructors; the same names resolve here
('swin_transformer', 'cvt_v4_transformer').  Keyword arguments (``is_teacher``, ``use_dense_prediction``) are forwarded
untouched, and an unknown name raises the reference's ValueError (its spelling included) so callers' error handling keeps
working.
"""
from . import registry


def build_model(config, **kwargs):
    name = config.MODEL.NAME
    try:
        factory = registry.model_entrypoints(name)
    except KeyError:
        raise ValueError('Unkown model: %s' % name) from None
    return factory(config, **kwargs)
