"""Trainer helpers: EMA, grad switches, latent sampling, mini-batch chunking.

Same names and semantics as the reference's src/gan_control/trainers/utils.py (cited per function), written
for device-resident execution: the EMA is two foreach kernels over all parameters instead of a Python loop of
per-tensor ops, and latent sampling takes an explicit device generator so every rank draws its own stream.
"""
import random
import weakref

import torch

# (name, parameter) lists per network, built once: Module.named_parameters() walks the whole module tree on every call, and one training
# iteration asks for it ~20 times (freeze / unfreeze, zero_grad, EMA, the None-gradient bookkeeping) -- ~3 ms of host time per step at
# FFHQ-1024 (tools/host_profile.py iter).  The parameter SET of a network does not change while it trains; a network whose set does change
# (load_state_dict(assign=True), added modules) is re-listed because the count no longer matches.
_NAMED = weakref.WeakKeyDictionary()
_HOOKED = weakref.WeakSet()  # networks whose load_state_dict already drops their entry


_FULL_CHECK_EVERY = 256      # calls between two complete re-listings (~13 training iterations)


def named_params(model):
    """list(model.named_parameters()), cached.  Validated in O(1) per call: the first and the last parameter must still be the objects
    registered under their names (a swapped parameter object -- load_state_dict(assign=True), .to() on a meta model -- re-lists).  A parameter
    or submodule replaced in the MIDDLE of the list is caught by the complete re-listing every _FULL_CHECK_EVERY calls and by the
    load_state_dict hook installed on first use; call forget_params(model) after adding or removing modules by hand.  The cache holds the
    owning modules weakly (an entry must not keep its own key -- the root module -- alive)."""
    hit = _NAMED.get(model)
    if hit is not None:
        lst, probes, calls = hit
        calls[0] += 1
        if calls[0] % _FULL_CHECK_EVERY == 0:
            fresh = list(model.named_parameters())
            if len(fresh) != len(lst) or any(a[1] is not b[1] or a[0] != b[0] for a, b in zip(fresh, lst)):
                hit = None
        if hit is not None:
            for owner, attr, p in probes:
                owner = owner()
                if owner is None or owner._parameters.get(attr) is not p:
                    hit = None
                    break
    if hit is None:
        lst = list(model.named_parameters())
        probes = []
        if lst:
            mods = dict(model.named_modules())
            for name, p in (lst[0], lst[-1]):
                owner, _, attr = name.rpartition('.')
                probes.append((weakref.ref(mods[owner]), attr, p))
        _NAMED[model] = (lst, probes, [0])
        if model not in _HOOKED and hasattr(model, 'register_load_state_dict_post_hook'):
            _HOOKED.add(model)
            model.register_load_state_dict_post_hook(lambda module, incompatible: forget_params(module))
    return lst


def forget_params(model):
    _NAMED.pop(model, None)


def zero_grad_none(model):
    """model.zero_grad(set_to_none=True) without the module-tree walk."""
    for _, p in named_params(model):
        p.grad = None


def accumulate(model1, model2, decay=0.999):
    """model1 <- decay * model1 + (1 - decay) * model2 over NAMED PARAMETERS only; buffers (the per-layer noise
    maps, blur kernels) are not averaged (reference utils.py:8-12)."""
    target = dict(named_params(model1))
    source = dict(named_params(model2))
    names = sorted(target)
    dst = [target[n].data for n in names]
    src = [source[n].data for n in names]
    torch._foreach_mul_(dst, decay)
    torch._foreach_add_(dst, src, alpha=1 - decay)
    from ..models.op import weight_cache
    weight_cache.invalidate(target.values())       # ``.data`` writes do not bump the version counters the cache of derived weight forms is keyed on


def requires_grad(model, flag=True):
    """Freeze / unfreeze every parameter of a network (reference utils.py:14-16)."""
    for _, param in named_params(model):
        param.requires_grad_(flag)


def make_noise(batch, latent_dim, n_noise, device, generator=None):
    """One [batch, latent_dim] normal sample, or a tuple of n_noise of them (reference utils.py:26-30)."""
    if n_noise == 1:
        return torch.randn(batch, latent_dim, device=device, generator=generator)
    stacked = torch.randn(n_noise, batch, latent_dim, device=device, generator=generator)
    return stacked.unbind(0)


def mixing_noise(batch, latent_dim, prob, device, generator=None):
    """Latents for one step: with probability ``prob`` two codes (style mixing), else a single one wrapped in a
    list (reference utils.py:19-23).  The coin uses Python's ``random`` like the reference."""
    two_styles = prob > 0 and random.random() < prob
    if two_styles:
        return make_noise(batch, latent_dim, 2, device, generator)
    return [make_noise(batch, latent_dim, 1, device, generator)]


def make_mini_batch_from_noise(noise, batch, mini_batch):
    """Split every latent tensor into batch // mini_batch chunks and regroup per chunk:
    [n_noise][batch, D] -> [n_chunks][n_noise][mini_batch, D] (reference utils.py:33-42)."""
    n_chunks = batch // mini_batch
    per_noise = [t.chunk(n_chunks) for t in noise]
    return [[pieces[c] for pieces in per_noise] for c in range(len(per_noise[0]))]


def set_grad_none(model, targets):
    """Drop the gradients of the named parameters so Adam skips them (reference utils.py:45-48)."""
    for name, param in named_params(model):
        if name in targets:
            param.grad = None
