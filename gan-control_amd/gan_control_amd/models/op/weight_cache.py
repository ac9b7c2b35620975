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

Batched refill: a parameter registered as part of a MODULE (``register(module)``) shares a group with the module's other parameters.  The
cache remembers, per parameter, which derived forms were asked for and how they are made (``recipe``); when the first form of a new
weight version is requested it recomputes ALL remembered forms of ALL parameters of the group with the backend's grouped kernels
(gc_weight_layout_grouped_f32, gc_conv2d_pack_weights_bf16x3_grouped): a handful of launches per network and optimiser step instead of
~4 per layer.  The grouped kernels share their per-element arithmetic with the single-tensor ones: bit-identical values.
``GANCONTROL_WEIGHT_BATCH=0`` turns the batching off.

Everything else -- an EMA network updated by somebody else's ``par.data.mul_().add_()`` (the reference's own ``accumulate``,
trainers/utils.py:8-12), a leaf tensor a caller passes in -- is recomputed on every call.  A root whose storage moved
(``module.to()``, ``load_state_dict(assign=True)``) is dropped on sight.
"""
import os
import warnings
import weakref

import torch

from ... import _lib

ENABLED = os.environ.get('GANCONTROL_WEIGHT_CACHE', '1') != '0'
BATCHED = os.environ.get('GANCONTROL_WEIGHT_BATCH', '1') != '0'

_roots = {}      # id(root) -> [weakref(root), version, {key: tensor}, data_ptr of the root]
_managed = {}    # id(tensor) -> weakref: roots with a known invalidation channel (see the module docstring)
_derived = {}    # data_ptr of a cached tensor -> (id(root), key)
_group_of = {}   # id(root) -> group id (parameters registered together as one module)
_group_members = {}   # group id -> [weakref(root)]
_recipes = {}    # id(root) -> [weakref(root), {key: recipe}]: every derived form ever asked of this root and how to make it (survives invalidation).  The weak
                 # reference pins the OWNER: Python re-uses the id of a dead tensor, and a parameter of a later network must not inherit the recipes (shapes!) of
                 # an unrelated dead one (round 6: `shape '[6, 5, 3, 3]' is invalid for input of size 1` in a long test session)
_warned = set()  # kinds of batched refill already reported as unsupported
stats = {'hit': 0, 'miss': 0, 'bypass': 0, 'batched': 0, 'batch_failed': 0}
batch_runner = None   # set by op/_backend.py: () -> callable(kind, items) -> [tensor] of the active backend, or None
pack_wanted = None    # set by op/_backend.py: () -> bool, does the active convolution arithmetic read hi / lo packs at all


generation = [0]      # bumped whenever anything is dropped: lets callers keep their own short-cuts over several cached tensors honest


def _drop(rid):
    generation[0] += 1
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
        _ensure_entry(root)
        return (rid, (src.storage_offset(), tuple(src.shape)))
    return None


def _ensure_entry(root):
    """The (fresh or still valid) cache entry of a managed root."""
    rid = id(root)
    entry = _roots.get(rid)
    if entry is None or entry[0]() is not root:
        _drop(rid)
        _roots[rid] = entry = [weakref.ref(root, lambda _, rid=rid: _forget(rid)), root._version, {}, root.data_ptr()]
    elif entry[1] != root._version or entry[3] != root.data_ptr():
        for t in entry[2].values():
            _derived.pop(t.data_ptr(), None)
        entry[1], entry[2], entry[3] = root._version, {}, root.data_ptr()
    return entry


def _recipes_of(root, create=False):
    """The recipe table of THIS tensor (not of a dead one whose id it re-uses)."""
    rid = id(root)
    ent = _recipes.get(rid)
    if ent is not None and ent[0]() is not root:
        _recipes.pop(rid, None)
        ent = None
    if ent is None:
        if not create or root is None:
            return {}
        ent = _recipes[rid] = [weakref.ref(root), {}]
    return ent[1]


def _forget(rid):
    _drop(rid)
    _recipes.pop(rid, None)
    _group_of.pop(rid, None)


def _refill_group(gid):
    """Recompute every remembered form of every parameter of group ``gid`` that is missing, level by level (forms of the parameter, forms of
    those forms, ...), each level's re-layouts in one grouped launch and its packs in another."""
    run = batch_runner() if batch_runner is not None else None
    if run is None:
        return
    todo = []
    for ref in _group_members.get(gid, ()):
        root = ref()
        if root is None or root.numel() == 0 or not root.is_contiguous():
            continue
        m = _managed.get(id(root))
        if m is None or m() is not root:
            continue
        bucket = _ensure_entry(root)[2]
        for key, recipe in _recipes_of(root).items():
            if key not in bucket:
                todo.append((len(key), id(root), root, key, recipe))
    if len(todo) < 2:
        return
    for depth in sorted({t[0] for t in todo}):
        level = [t for t in todo if t[0] == depth]
        for kind in ('layout', 'pack', 'wsq'):
            items, owners = [], []
            for _, rid, root, key, recipe in level:
                if recipe[0] != kind:
                    continue
                bucket = _roots[rid][2]
                if len(key) == 3:                       # parent = a contiguous view of the parameter: (storage offset, shape)
                    off, shape = key[0], key[1]
                    n = 1
                    for d in shape:
                        n *= d
                    src = root.detach().reshape(-1)[off - root.storage_offset(): off - root.storage_offset() + n].view(shape)
                else:
                    src = bucket.get(key[:-1])
                    if src is None:
                        continue                        # its parent could not be made in this pass: the caller's own make() will do it
                items.append((src,) + tuple(recipe[1:]))
                owners.append((rid, key))
            if not items or (kind == 'pack' and pack_wanted is not None and not pack_wanted()):
                continue            # (packs recorded under a split-bf16 mode are not rebuilt while the exact-fp32 mode is active)
            try:
                outs = run(kind, items)
            except _lib.UnsupportedError as e:
                # one item the grouped entry cannot take (a pack whose descriptor has no packed form any more, a table limit) must not abort
                # the forward pass: leave this kind to the callers' own make(), which handles each tensor on its own.  Only the library's
                # "unsupported" answer is taken this way: a HIP failure, a bad descriptor or a host-side error is raised where it happened
                stats['batch_failed'] += 1
                if kind not in _warned:
                    _warned.add(kind)
                    warnings.warn('gan_control_amd: batched refill of the %r weight forms is unsupported for this network (%s); '
                                  'falling back to one launch per tensor' % (kind, e))
                continue
            for (rid, key), out in zip(owners, outs):
                if out is not None and out.numel() > 0 and out.data_ptr() not in _derived:
                    _roots[rid][2][key] = out
                    _derived[out.data_ptr()] = (rid, key)
                    stats['batched'] += 1



def derive(src, op, make, recipe=None):
    """``make()`` -- a tensor computed from ``src`` alone by the operation named ``op`` (hashable) -- cached for as
    long as the root of ``src`` keeps its version.  Returns the cached tensor itself: callers must not modify it, and an
    autograd Function must return an alias (``.detach()``), not this object.

    ``recipe`` = ('layout', taps, k, n, src_stride, dst_shape, dst_stride, flip, scale) | ('pack', conv-desc fields) | ('wsq',): how the backend's
    grouped kernels make the same tensor, which lets a miss refill the whole parameter group at once (module docstring)."""
    where = _root_of(src) if ENABLED else None
    if where is None:
        stats['bypass'] += 1
        return make()
    rid, prefix = where
    key = prefix + (op,)
    bucket = _roots[rid][2]
    out = bucket.get(key)
    if out is None and recipe is not None:
        if callable(recipe):             # built on a miss only: the hot path (a hit) pays nothing for it
            recipe = recipe()
        owner = _roots[rid][0]()
        _recipes_of(owner, create=True)[key] = recipe
        gid = _group_of.get(rid)
        if gid is not None and not any(ref() is owner for ref in _group_members.get(gid, ())):
            _group_of.pop(rid, None)         # a stale mapping of a dead tensor whose id this one re-uses
            gid = None
        if BATCHED and gid is not None:
            _refill_group(gid)
            bucket = _roots[rid][2]
            out = bucket.get(key)
            if out is not None:
                return out
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
    gid = None
    if isinstance(obj, torch.nn.Module):
        gid = id(obj)
        obj = list(obj.parameters())
        _group_members[gid] = [weakref.ref(t) for t in obj]
    elif torch.is_tensor(obj):
        obj = (obj,)
    for t in obj:
        tid = id(t)
        m = _managed.get(tid)
        if m is None or m() is not t:
            _managed[tid] = weakref.ref(t, lambda _, tid=tid: _managed.pop(tid, None))
        if gid is not None:
            _group_of[tid] = gid


def unregister_all():
    """Forget every registration and every cached tensor (tests)."""
    _managed.clear()
    _group_of.clear()
    _group_members.clear()
    _recipes.clear()
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
