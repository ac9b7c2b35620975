"""Host-side logic of the operator socket, validated on the CPU over the emulated C ABI.

These tests exercise the PRODUCT's autograd layer (gan_control_amd.models.op and the modules
in gan_control_amd.models.gan_model): gradient closure to second order, geometry, None-vs-zero
gradient structure.  The kernels themselves are covered by the ``-m gpu`` tests.
"""
import pytest
import torch

import op_checks as oc


@pytest.mark.parametrize('case', oc.names(oc.UPF))
def test_upfirdn2d(case, emu_backend):
    oc.check_upfirdn2d(case, 'cpu')


@pytest.mark.parametrize('case', oc.names(oc.BA))
def test_bias_act(case, emu_backend):
    oc.check_bias_act(case, 'cpu')


def test_noise_bias_act(emu_backend):
    oc.check_noise_bias_act('cpu')


@pytest.mark.parametrize('case', oc.names(oc.CV, 'conv_'))
def test_equal_conv(case, emu_backend):
    oc.check_equal_conv(case, 'cpu')


@pytest.mark.parametrize('case', oc.names(oc.CV, 'mod_'))
def test_modulated_conv(case, emu_backend):
    oc.check_modulated_conv(case, 'cpu')


def test_conv_functional(emu_backend):
    oc.conv_functional_checks('cpu')


@pytest.mark.parametrize('size', [32])
def test_network(size, emu_backend):
    oc.check_network(size, 'cpu')


@pytest.mark.parametrize('precision', ['high', 'medium'])
def test_generator_twice_under_non_default_matmul_precision(precision, emu_backend):
    """the fused style path's latent gather must stay exact, and must not touch process state, under a non-default
    set_float32_matmul_precision -- the second forward used to die inside torch's legacy / new API check.  (Image parity is not asserted
    here: on the CPU 'medium' also lowers the emulated backend's own ATen products.)"""
    from gan_control_amd.models import gan_model as gm
    before = torch.get_float32_matmul_precision()
    torch.set_float32_matmul_precision(precision)
    try:
        g, _ = oc.build_models(32, 'cpu')
        z = torch.randn(2, 512, generator=torch.Generator().manual_seed(3))
        for _ in range(2):
            img, lat = g([z], return_latents=True)
            assert torch.isfinite(img).all()
            assert torch.get_float32_matmul_precision() == precision
        plan = gm._STYLE_PLANS[g]
        lat = lat.detach().clone().requires_grad_(True)
        rows = gm._gather_latents(plan, lat)
        want = lat.transpose(0, 1)[torch.tensor(plan.latent_index)].reshape(len(plan.latent_index), -1)
        assert torch.equal(rows, want)
        rows.backward(torch.ones_like(rows))
        counts = torch.bincount(torch.tensor(plan.latent_index), minlength=lat.shape[1]).float()
        assert torch.equal(lat.grad, counts[None, :, None].expand_as(lat))
    finally:
        torch.set_float32_matmul_precision(before)


def test_no_cpu_fallback():
    """Without the emulation the product refuses CPU tensors instead of silently falling back."""
    from gan_control_amd.models.op import upfirdn2d, fused_leaky_relu
    with pytest.raises(RuntimeError):
        upfirdn2d(torch.zeros(1, 1, 4, 4), torch.ones(2, 2))
    with pytest.raises(RuntimeError):
        fused_leaky_relu(torch.zeros(1, 2, 4, 4), torch.zeros(2))


def test_split_fc(emu_backend):
    oc.check_split_fc('cpu')


def test_mixing_truncation_and_stored_noise(emu_backend):
    oc.check_mixing_truncation('cpu')


def test_noise_modes_zeros_and_id_zeros(emu_backend):
    oc.check_noise_modes('cpu')


def test_transfer_learning_load(emu_backend):
    oc.check_transfer_learning('cpu')


def test_misc_modules(emu_backend):
    oc.check_misc('cpu')


def _fused_step_conv_check(device):
    """A FUSED optimiser step and a ``.data`` write do not bump the autograd version counter the cache of derived weight forms is keyed
    on: the next forward must still see the new weights (regression: the cache served the weights of the first iteration for ever)."""
    from gan_control_amd.models.op import conv2d_gradfix, weight_cache
    from gan_control_amd.trainers.utils import accumulate
    weight_cache.clear()
    gen = torch.Generator().manual_seed(3)
    conv = torch.nn.Conv2d(5, 6, 3, bias=False).to(device)                  # only a holder of a [6, 5, 3, 3] parameter
    with torch.no_grad():
        conv.weight.copy_(torch.randn(6, 5, 3, 3, generator=gen))
    x = torch.randn(2, 5, 8, 8, generator=gen).to(device)
    fwd = lambda w: conv2d_gradfix.conv2d(x, w, padding=1, weight_scale=0.5)
    opt = torch.optim.Adam(conv.parameters(), lr=0.1, fused=True)
    fwd(conv.weight).square().sum().backward()
    version = conv.weight._version
    opt.step()
    y_cached = fwd(conv.weight).detach()
    weight_cache.ENABLED, prev = False, weight_cache.ENABLED
    try:
        y_fresh = fwd(conv.weight).detach()
    finally:
        weight_cache.ENABLED = prev
    assert torch.equal(y_cached, y_fresh), 'stale derived weights after a fused optimiser step (version counter %d -> %d)' % (version, conv.weight._version)
    assert weight_cache.stats['hit'] + weight_cache.stats['miss'] > 0, 'a parameter an optimiser has stepped must be cached'
    # the EMA writes through .data (trainers/utils.py::accumulate, like the reference's): same requirement.  Registered (this
    # package's trainer does that; its accumulate() invalidates) and NOT registered (somebody else's training loop, the
    # reference's own accumulate(): `par.data.mul_().add_()`, trainers/utils.py:8-12 -- such a tensor must never be cached).
    for registered in (True, False):
        ema = torch.nn.Conv2d(5, 6, 3, bias=False).to(device)
        if registered:
            weight_cache.register(ema)
        before = fwd(ema.weight).detach()
        bypass0 = weight_cache.stats['bypass']
        if registered:
            accumulate(ema, conv, 0.5)
        else:
            for p_ema, p in zip(ema.parameters(), conv.parameters()):
                p_ema.data.mul_(0.5).add_(p.data, alpha=0.5)
        after = fwd(ema.weight).detach()
        assert registered or weight_cache.stats['bypass'] > bypass0, 'an unmanaged tensor was cached'
        weight_cache.ENABLED, prev = False, weight_cache.ENABLED
        try:
            fresh = fwd(ema.weight).detach()
        finally:
            weight_cache.ENABLED = prev
        assert torch.equal(after, fresh) and not torch.equal(after, before)
    # a parameter whose storage was swapped (module.to(), load_state_dict(assign=True)) is dropped on sight
    weight_cache.register(conv)
    y0 = fwd(conv.weight).detach()
    conv.weight.data = (conv.weight.data * 3.0).clone()
    weight_cache.ENABLED, prev = False, weight_cache.ENABLED
    try:
        fresh = fwd(conv.weight).detach()
    finally:
        weight_cache.ENABLED = prev
    assert torch.equal(fwd(conv.weight).detach(), fresh) and not torch.equal(fresh, y0)
    weight_cache.clear()


def test_weight_cache_survives_fused_optimizer_and_data_writes(emu_backend):
    _fused_step_conv_check('cpu')


def test_training_with_and_without_weight_cache(emu_backend):
    """Three full iterations (fused Adam, EMA) with the cache of derived weight forms on (batched refill of a whole network's forms per
    optimiser step), on without batching, and off: the same parameters, bit for bit."""
    from gan_control_amd.models.op import weight_cache
    out = []
    for enabled, batched in ((True, True), (True, False), (False, False)):
        weight_cache.clear()
        weight_cache.ENABLED, prev = enabled, weight_cache.ENABLED
        weight_cache.BATCHED, prev_b = batched, weight_cache.BATCHED
        before = weight_cache.stats['batched']
        try:
            from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config
            tr = GeneratorTrainer(default_config(16, 4), device='cpu', seed=0, fused_adam=True)
            real = tr.synthetic_batch()
            for i in range(3):
                tr.train_iteration(i, real)
            out.append({k: v.clone() for k, v in list(tr.generator.state_dict().items()) + list(tr.discriminator.state_dict().items())})
            assert (weight_cache.stats['batched'] > before) == batched, 'batched refill %s' % ('did not run' if batched else 'ran although switched off')
        finally:
            weight_cache.ENABLED, weight_cache.BATCHED = prev, prev_b
    for k in out[0]:
        assert torch.equal(out[0][k], out[1][k]) and torch.equal(out[0][k], out[2][k]), k
    weight_cache.clear()


@pytest.mark.parametrize('name', ['ffhq', 'metfaces', 'afhq'])
def test_trainer_from_shipped_config(name, emu_backend):
    oc.check_config_ingestion('cpu', name, size=16, batch=4)


def test_weight_cache_reuse_and_invalidation(emu_backend):
    """Derived weight forms (kernel layout, its adjoint) are computed once per weight version: reused by the next call, recomputed
    after an in-place update, and never change a result or a gradient."""
    from gan_control_amd.models.op import conv2d_gradfix, weight_cache
    weight_cache.clear()
    gen = torch.Generator().manual_seed(0)
    w = torch.nn.Parameter(torch.randn(6, 5, 3, 3, generator=gen))
    weight_cache.register(w)
    x = torch.randn(2, 5, 8, 8, generator=gen, requires_grad=True)

    def run():
        y = conv2d_gradfix.conv2d(x, w, padding=1, weight_scale=0.5)
        gx, gw = torch.autograd.grad(y.square().sum(), [x, w])
        return y.detach(), gx, gw

    before = dict(weight_cache.stats)
    a = run()
    first = {k: weight_cache.stats[k] - before[k] for k in before}
    b = run()
    second = {k: weight_cache.stats[k] - before[k] - first[k] for k in before}
    assert first['miss'] == 2 and first['hit'] == 0, first            # kernel layout + its adjoint (input-gradient weights)
    assert second['miss'] == 0 and second['hit'] == 2, second
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    with torch.no_grad():
        w.mul_(2.0)                                                      # what an optimiser step does: bumps the version counter
    c = run()
    assert weight_cache.stats['miss'] - before['miss'] == 4
    assert torch.allclose(c[0], 2 * a[0], rtol=1e-6) and torch.allclose(c[2], 2 * a[2], rtol=1e-5)
    weight_cache.ENABLED, prev = False, weight_cache.ENABLED
    try:
        d = run()
    finally:
        weight_cache.ENABLED = prev
    for u, v in zip(c, d):
        assert torch.equal(u, v)
    del w
    import gc
    gc.collect()
    weight_cache.clear()


def test_resblock_blur_adjoint_fusion(emu_backend):
    oc.check_resblock_blur_adjoint_fusion('cpu', size=64, batch=2)


@pytest.mark.parametrize('level', [1, 2])
def test_scale_grads_from_sample_wgrad(level, emu_backend):
    oc.check_scale_grads_from_sample_wgrad('cpu', size=32, batch=2, tol=1e-10, dtype=torch.float64, level=level)


def test_torgb_fork(emu_backend):
    oc.check_torgb_fork('cpu', size=32, batch=2, tol=1e-10, dtype=torch.float64)


def test_weight_cache_recipes_do_not_outlive_their_tensor():
    """Python re-uses the id of a dead tensor: a later parameter must not inherit the remembered derived forms (with their shapes) of an unrelated dead one
    (round 6: a long test session hit `shape '[6, 5, 3, 3]' is invalid for input of size 1` inside the batched refill)."""
    import weakref
    import torch
    from gan_control_amd.models.op import weight_cache as wc
    dead = torch.nn.Parameter(torch.zeros(6, 5, 3, 3))
    live = torch.nn.Parameter(torch.zeros(1))
    stale = [weakref.ref(dead), {('layout',): ('layout', 9, 5, 6)}]
    wc._recipes[id(live)] = stale                      # what an id collision leaves behind
    try:
        assert wc._recipes_of(live) == {} and id(live) not in wc._recipes
        mine = wc._recipes_of(live, create=True)
        mine[('wsq',)] = ('wsq',)
        assert wc._recipes_of(live) == {('wsq',): ('wsq',)} and wc._recipes[id(live)][0]() is live
    finally:
        wc._recipes.pop(id(live), None)
