"""Activation-dtype caches of fp32 parameters (the GEMM kernels read weights in the activation dtype).

A cached copy is refreshed when the parameter's autograd version changes (torch optimizers, load_state_dict)
or when ``invalidate()`` is called (the fused HIP update writes parameters through raw pointers, which does
not bump the version counter)."""
import torch

from . import ops

_GEN = 0
_CACHE = {}
_MANAGED = set()   # ids of frozen parameters whose every update goes through the fused updater (the EMA teacher)


def manage(p):
    """Declare that `p` (requires_grad=False) is only ever written by the fused update kernel, which refreshes its
    cached cast itself; without this, frozen parameters are recast on every use because in-place `.data` updates
    (main_esvit.py:590) are invisible to version counters."""
    _MANAGED.add(id(p))


def is_managed(p):
    return id(p) in _MANAGED


def invalidate():
    global _GEN
    _GEN += 1


def _tag(p):
    return (p._version, _GEN, p.data_ptr(), ops.act_dtype())


def cached_cast(p, shape2d=None):
    """fp32 parameter -> activation-dtype copy (optionally viewed as a 2-D matrix first)."""
    key = (id(p), "cast")
    tag = _tag(p)
    ent = _CACHE.get(key)
    if ent is not None and ent[0] == tag:
        return ent[1]
    src = p.detach()
    if shape2d is not None:
        src = src.reshape(shape2d)
    w = ops.cast_to_act(src.contiguous())
    _CACHE[key] = (tag, w)
    return w


def cached(p, name, fn):
    """generic per-parameter cache for derived tensors (e.g. the weight-normed last layer)."""
    key = (id(p), name)
    tag = _tag(p)
    ent = _CACHE.get(key)
    if ent is not None and ent[0] == tag:
        return ent[1]
    val = fn()
    _CACHE[key] = (tag, val)
    return val


def cast_buffer_ptr(p, fresh):
    """Device pointer of p's cached bf16 cast (0 if there is none, or the activation dtype is not bf16).  The fused
    update writes the new values straight into that buffer; the caller passes the collected `fresh` list to
    mark_fresh() after invalidate() so the entries are not recast."""
    if ops.act_dtype() != torch.bfloat16:
        return 0
    key = (id(p), "cast")
    ent = _CACHE.get(key)
    if ent is None or ent[1].dtype != torch.bfloat16 or ent[1].numel() != p.numel() or ent[1].data_ptr() == p.data_ptr():
        return 0
    fresh.append((key, p))
    return ent[1].data_ptr()


def mark_fresh(fresh):
    for key, p in fresh:
        ent = _CACHE.get(key)
        if ent is not None:
            _CACHE[key] = (_tag(p), ent[1])


def clear():
    _CACHE.clear()
