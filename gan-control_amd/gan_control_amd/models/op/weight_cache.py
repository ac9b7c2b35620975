"""Derived forms of a weight tensor, computed once per optimiser step instead of once per call.

A convolution weight enters the kernels as ``w_t`` (kernel layout, scaled), as ``adj(w_t)`` (input-gradient
weights) and -- in split-bf16 mode -- as the hi / lo packs of both.  The reference recomputes ``weight * scale`` and
the layout permutes of ``aten::convolution_backward`` on every call (gan_model.py:154, 284, 295-303); the parameter,
however, only changes when the optimiser steps, while one training iteration uses it 2-4 times (D step: forward +
input gradient over the interleaved fake / real batch; R1; the G forward of the D step and of the G step, ...).

The cache is keyed on the ROOT tensor (the parameter, or the leaf a test passes in) and its autograd version counter
(in-place updates through the dispatcher -- ``copy_``, ``load_state_dict``, foreach / single-tensor optimisers -- bump it).
The version counter alone is NOT enough: the FUSED optimisers (``Adam(fused=True)``, the default on the device) update parameters
without bumping it, and so does anything that writes through ``.data`` (the EMA of trainers/utils.py, the reference's own
``accumulate``).  Every ``Optimizer.step`` therefore drops what was derived from that optimiser's parameters (a global post-step hook registered below), and
``invalidate()`` is what code that writes through ``.data`` must call (``accumulate`` does).  Derived tensors
register their storage address, which lets a derivation of a derivation (``adj(w_t)``, ``pack(adj(w_t))``) find its
root without any attribute travelling through ``save_for_backward``.  The cache owns the derived tensors while they
are valid, so a registered address cannot be re-used by another live tensor.  Results are bit-identical to
recomputing (same kernels, same inputs).  ``GANCONTROL_WEIGHT_CACHE=0`` turns it off.

Which roots are cached at all: only tensors with a KNOWN invalidation channel --

* parameters an optimiser has stepped (the post-step hook below both invalidates them and marks them as managed: from the
  second iteration on every trained parameter is cached, whoever wrote the training loop), and
* parameters handed to ``register()`` by code that promises to call ``invalidate()`` after writing them behind the version
  counter (this package's trainer registers G, D and the EMA copy; its ``accumulate`` invalidates the EMA).

Everything else -- an EMA network updated by somebody else's ``par.data.mul_().add_()`` (the reference's own ``accumulate``,
trainers/utils.py:8-12), a leaf tensor a caller passes in -- is recomputed on every call.  A root whose storage moved
(``module.to()``, ``load_state_dict(assign=True)``) is dropped on sight.
"""
import os
import weakref

import torch

ENABLED = os.environ.get('GANCONTROL_WEIGHT_CACHE', '1') != '0'

_roots = {}      # id(root) -> [weakref(root), version, {key: tensor}, data_ptr of the root]
_managed = {}    # id(tensor) -> weakref: roots with a known invalidation channel (see the module docstring)
_derived = {}    # data_ptr of a cached tensor -> (id(root), key)
stats = {'hit': 0, 'miss': 0, 'bypass': 0}


def _drop(rid):
    entry = _roots.pop(rid, None)
    if entry is not None:
        for t in entry[2].values():
            _derived.pop(t.data_ptr(), None)


def _root_of(src):
    """(root id, key prefix) of a tensor the cache may derive from, or None."""
    if src.numel() == 0:
        return None
    hit = _derived.get(src.data_ptr())
    if hit is not None:
        entry = _roots.get(hit[0])
        if entry is not None and entry[0]() is not None and entry[0]()._version == entry[1] and hit[1] in entry[2] \
                and entry[2][hit[1]].shape == src.shape:
            return hit
    root = src._base if src._is_view() and src._base is not None else src
    if root.grad_fn is None and (isinstance(root, torch.nn.Parameter) or root.requires_grad) and root.is_contiguous() and src.is_contiguous():
        rid = id(root)
        m = _managed.get(rid)
        if m is None or m() is not root:
            return None                   # nobody promised to tell the cache when this tensor changes: recompute
        entry = _roots.get(rid)
        if entry is None or entry[0]() is not root:
            _drop(rid)
            _roots[rid] = entry = [weakref.ref(root, lambda _, rid=rid: _drop(rid)), root._version, {}, root.data_ptr()]
        elif entry[1] != root._version or entry[3] != root.data_ptr():
            for t in entry[2].values():
                _derived.pop(t.data_ptr(), None)
            entry[1], entry[2], entry[3] = root._version, {}, root.data_ptr()
        return (rid, (src.storage_offset(), tuple(src.shape)))
    return None


def derive(src, op, make):
    """``make()`` -- a tensor computed from ``src`` alone by the operation named ``op`` (hashable) -- cached for as
    long as the root of ``src`` keeps its version.  Returns the cached tensor itself: callers must not modify it, and an
    autograd Function must return an alias (``.detach()``), not this object."""
    where = _root_of(src) if ENABLED else None
    if where is None:
        stats['bypass'] += 1
        return make()
    rid, prefix = where
    key = prefix + (op,)
    bucket = _roots[rid][2]
    out = bucket.get(key)
    if out is None:
        stats['miss'] += 1
        out = make()
        if out.numel() > 0 and out.data_ptr() not in _derived:
            bucket[key] = out
            _derived[out.data_ptr()] = (rid, key)
    else:
        stats['hit'] += 1
    return out


def clear():
    for rid in list(_roots):
        _drop(rid)


def register(obj):
    """Mark parameters as cacheable: ``obj`` is a module, a tensor or an iterable of tensors.  The caller promises that every write
    that does not go through an optimiser step or bump the autograd version counter (``.data`` arithmetic, fused kernels of its own,
    collectives on ``.data``) is followed by ``invalidate()`` on the written tensors."""
    if isinstance(obj, torch.nn.Module):
        obj = obj.parameters()
    elif torch.is_tensor(obj):
        obj = (obj,)
    for t in obj:
        tid = id(t)
        m = _managed.get(tid)
        if m is None or m() is not t:
            _managed[tid] = weakref.ref(t, lambda _, tid=tid: _managed.pop(tid, None))


def unregister_all():
    """Forget every registration and every cached tensor (tests)."""
    _managed.clear()
    clear()


def invalidate(tensors=None):
    """Drop what was derived from ``tensors`` (parameters whose storage was written behind the version counter's back: ``.data`` writes,
    fused optimiser kernels); everything when called without arguments."""
    if tensors is None:
        return clear()
    for t in tensors:
        _drop(id(t))


def _after_optimizer_step(optimizer, args, kwargs):
    # only the parameters this optimiser owns: the other network's derived forms stay valid across this step
    for group in optimizer.param_groups:
        invalidate(group['params'])
        register(group['params'])          # an optimiser owns them: this hook is their invalidation channel from now on


# every torch optimiser, whichever implementation (fused kernels do not touch the version counters)
from torch.optim.optimizer import register_optimizer_step_post_hook as _register      # noqa: E402
_register(_after_optimizer_step)
