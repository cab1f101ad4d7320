"""Activation-dtype caches of fp32 parameters (the GEMM kernels read weights in the activation dtype) and the
gradient sink of the data-parallel reducer.

A cached copy is refreshed when the parameter's autograd version changes (torch optimizers, load_state_dict)
or when ``invalidate()`` is called (the fused HIP update writes parameters through raw pointers, which does
not bump the version counter).  Entries are keyed by ``id(p)`` but hold a weak reference to the parameter and are
ignored (and dropped) when it has died, so a new model whose tensors reuse the ids / addresses of a freed one never
sees its stale casts."""
import weakref

import torch

from . import ops

_GEN = 0
_CACHE = {}        # (id(p), name) -> (weakref(p), tag, value)
_MANAGED = {}      # id(p) -> weakref(p): frozen parameters whose every update goes through the fused updater (the EMA teacher)
_GRAD_SINK = None  # id(p) -> preallocated fp32 gradient tensor (a bucket view of engine.GradBucketReducer), or None
_GRAD_TAKEN = set()  # ids whose slot has been handed to a producer during the current backward


def _drop(pid):
    for key in [k for k in _CACHE if k[0] == pid]:
        _CACHE.pop(key, None)
    _MANAGED.pop(pid, None)


def _ref(p):
    pid = id(p)
    return weakref.ref(p, lambda _r, pid=pid: _drop(pid))


def manage(p):
    """Declare that `p` (requires_grad=False) is only ever written by the fused update kernel, which refreshes its
    cached cast itself; without this, frozen parameters are recast on every use because in-place `.data` updates
    (main_esvit.py:590) are invisible to version counters."""
    _MANAGED[id(p)] = _ref(p)


def is_managed(p):
    r = _MANAGED.get(id(p))
    return r is not None and r() is p


def invalidate():
    global _GEN
    _GEN += 1


def _tag(p):
    return (p._version, _GEN, p.data_ptr(), ops.act_dtype())


def _lookup(p, name):
    ent = _CACHE.get((id(p), name))
    if ent is not None and ent[0]() is p:
        return ent
    return None


def cached_cast(p, shape2d=None):
    """fp32 parameter -> activation-dtype copy (optionally viewed as a 2-D matrix first)."""
    tag = _tag(p)
    ent = _lookup(p, "cast")
    if ent is not None and ent[1] == tag:
        return ent[2]
    src = p.detach()
    if shape2d is not None:
        src = src.reshape(shape2d)
    w = ops.cast_to_act(src.contiguous())
    _CACHE[(id(p), "cast")] = (_ref(p), tag, w)
    return w


_PROBE_STALE = __import__("os").environ.get("ESVIT_PROBE_STALE_DERIVED", "0") == "1"  # timing probe ONLY (wrong results): derived copies never refreshed


def cached(p, name, fn):
    """generic per-parameter cache for derived tensors (e.g. the weight-normed last layer)."""
    tag = _tag(p)
    ent = _lookup(p, name)
    if ent is not None and (ent[1] == tag or (_PROBE_STALE and name.startswith(("MLP_", "ATTN_")))):
        return ent[2]
    val = fn()
    _CACHE[(id(p), name)] = (_ref(p), tag, val)
    return val


def cast_buffer_ptr(p, fresh):
    """Device pointer of p's cached bf16 cast (0 if there is none, or the activation dtype is not bf16).  The fused
    update writes the new values straight into that buffer; the caller passes the collected `fresh` list to
    mark_fresh() after invalidate() so the entries are not recast."""
    if ops.act_dtype() != torch.bfloat16:
        return 0
    ent = _lookup(p, "cast")
    if ent is None or ent[2].dtype != torch.bfloat16 or ent[2].numel() != p.numel() or ent[2].data_ptr() == p.data_ptr():
        return 0
    fresh.append(((id(p), "cast"), p))
    return ent[2].data_ptr()


def mark_fresh(fresh):
    for key, p in fresh:
        ent = _CACHE.get(key)
        if ent is not None and ent[0]() is p:
            _CACHE[key] = (ent[0], _tag(p), ent[2])


def clear():
    _CACHE.clear()


# ---- gradient sink ------------------------------------------------------------------------------------------------
def set_grad_sink(views):
    """views: dict id(parameter) -> fp32 tensor of the parameter's shape the weight-gradient GEMMs of the next backward
    write into (engine.GradBucketReducer's bucket slots), or None to switch the sink off."""
    global _GRAD_SINK
    _GRAD_SINK = views
    _GRAD_TAKEN.clear()


def grad_out(p, shape2d=None):
    """the preallocated gradient tensor of parameter `p` (viewed as `shape2d` if given), or None.

    A slot is handed out AT MOST ONCE per backward.  A schedule that uses a parameter several times in one graph (one
    backbone pass per resolution group, swin_transformer.py:729-751; CvT) gets the slot for its first contribution and
    None -- i.e. a fresh tensor -- for the others: autograd sums the contributions before AccumulateGrad runs, and the
    reducer's hook packs the sum if it did not end up in the slot.  (Handing the slot out twice would make the second
    producer overwrite the first and autograd add two aliases of the same memory.)"""
    if _GRAD_SINK is None or p is None:
        return None
    v = _GRAD_SINK.get(id(p))
    if v is None or id(p) in _GRAD_TAKEN:
        return None
    _GRAD_TAKEN.add(id(p))
    return v if shape2d is None else v.view(shape2d)
