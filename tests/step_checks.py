"""Step-level parity: the product trainer must reproduce the iteration captured from the reference."""
import torch

from conftest import load_golden, rel_err
import op_checks as oc


def make_trainer(device, size=32, batch=4):
    from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config
    from oracle.networks import procedural_fill_
    tr = GeneratorTrainer(default_config(size, batch), device=device, seed=0, fused_adam=False)
    tr.generator.load_state_dict(procedural_fill_(tr.generator.state_dict()))
    tr.discriminator.load_state_dict(procedural_fill_(tr.discriminator.state_dict()))
    tr.g_ema.load_state_dict(tr.generator.state_dict())
    return tr


def step_inputs(s, size, batch):
    """The iteration's inputs.  The 32x32 fixture stores them; the BASELINE-size fixtures store the seed, and the tensors are
    redrawn here in oracle/make_golden.py::golden_step's order (torch's CPU generator is bit-reproducible); the stored
    latents prove the redraw is the stream the reference run consumed."""
    gen = torch.Generator().manual_seed(int(s['input_seed']))
    real = torch.rand(batch, 3, size, size, generator=gen) * 2 - 1
    z_d, z_g = torch.randn(batch, 512, generator=gen), torch.randn(batch, 512, generator=gen)
    z_pl = torch.randn(batch // 2, 512, generator=gen)
    pl_noise = torch.randn(batch // 2, 3, size, size, generator=gen)
    for k, v in (('z_d', z_d), ('z_g', z_g), ('z_pl', z_pl)):
        assert torch.equal(v, torch.from_numpy(s[k])), 'the seeded redraw does not reproduce the fixture: ' + k
    if 'real' in s:
        assert torch.equal(real, torch.from_numpy(s['real'])) and torch.equal(pl_noise, torch.from_numpy(s['pl_noise']))
    return {'real': real, 'z_d': z_d, 'z_g': z_g, 'z_pl': z_pl, 'pl_noise': pl_noise}


def _check_grads(module, s, phase, stat, tol):
    """Gradients of one backward pass against the reference's: the set of parameters that have one, the global norm
    (``stat/<stat>``) and the norm of every single parameter's gradient (``gradnorm/<phase>/*``)."""
    got = {n: p.grad for n, p in module.named_parameters() if p.grad is not None}
    ref_total = float(s[f'stat/{stat}'])
    total = float(torch.stack([g.double().pow(2).sum() for g in got.values()]).sum().sqrt())
    assert abs(total - ref_total) <= tol * ref_total, (stat, total, ref_total)
    names = [str(n) for n in s[f'gradnorm/{phase}/names']]
    assert sorted(names) == sorted(got), (phase, set(names) ^ set(got))
    floor = 1e-4 * ref_total          # parameters whose gradient is at the rounding-noise level of the pass are bounded absolutely
    worst = 0.0
    scalars = []
    for n, ref in zip(names, s[f'gradnorm/{phase}/vals']):
        ref, val = float(ref), float(got[n].double().norm())
        if got[n].numel() == 1:
            # a NoiseInjection strength: ONE scalar = a sum over batch x channels x pixels of random-sign terms, cancelling to a
            # small fraction of their magnitude.  The sign-like first Adam steps of the earlier passes (an element whose
            # gradient is ~0 may step the other way) perturb it by ~1e-4 absolute whatever its size, so the 13-17 strengths
            # of the network are compared together as one vector instead of one relative error each.
            scalars.append((val, ref))
            continue
        err = abs(val - ref) / max(ref, floor)
        worst = max(worst, err)
        assert err <= tol, (phase, n, val, ref)
    if scalars:
        a, b = torch.tensor(scalars, dtype=torch.float64).unbind(1)
        err = float((a - b).norm() / b.norm().clamp_min(floor))
        worst = max(worst, err)
        assert err <= tol, (phase, 'scalar parameters', scalars)
    return worst


def check_step(device, tol=2e-3, name='step'):
    """One full iteration (D step, R1, G step, path-length, EMA) of the product trainer against the iteration captured from
    the reference: loss scalars, path lengths, the gradients of all four backward passes (global and per-parameter norms),
    the None-gradient name sets and sampled parameter values after the four Adam updates.  ``name`` selects the fixture:
    'step' (32x32, batch 4), 'step_512' (512x512, batch 4), 'step_1024' (1024x1024, batch 2)."""
    from gan_control_amd.trainers.utils import requires_grad, accumulate
    s = load_golden(name)
    size, batch = [int(v) for v in s['cfg']]
    inputs = step_inputs(s, size, batch)
    t = lambda k: (inputs[k] if k in inputs else torch.from_numpy(s[k])).to(device)
    tr = make_trainer(device, size=size, batch=batch)
    assert sorted(tr.none_g_grads) == sorted(str(n) for n in s['none_g'])
    assert sorted(tr.none_d_grads) == sorted(str(n) for n in s['none_d'])
    seeds = [int(v) for v in s['noise_seeds']]
    real = t('real')
    # iteration 0 with the fixture's latents and noise maps
    requires_grad(tr.generator, False); requires_grad(tr.discriminator, True)
    # Gradient tolerances: `tol` for the two plain backward passes, 3 * tol for the two double-backward passes (R1, path length).
    # In fp64 the product reproduces the oracle's double-backward gradients to 1e-15; in fp32 ONE leaky-ReLU whose pre-activation
    # is within rounding of zero (expected ~0.4 per million activations) takes the other slope and moves a whole layer's
    # second-order gradient by ~5e-4 (measured: tools/pl_error_probe.py), on top of the sign-like Adam steps that precede it.
    worst = {}
    tr.discriminator_step([[t('z_d')]], [real], noise=oc.seeded_noise(size, batch, seeds[0], device))
    worst['d'] = _check_grads(tr.discriminator, s, 'd', 'd_grad_norm', tol)
    tr.discriminator_regularize_step([real])
    worst['r1'] = _check_grads(tr.discriminator, s, 'r1', 'r1_grad_norm', 3 * tol)
    requires_grad(tr.generator, True); requires_grad(tr.discriminator, False)
    tr.generator_step([[t('z_g')]], noise=oc.seeded_noise(size, batch, seeds[1], device))
    worst['g'] = _check_grads(tr.generator, s, 'g', 'g_grad_norm', tol)
    tr.generator_regularize_step(noise=oc.seeded_noise(size, batch // 2, seeds[2], device), pl_noise=t('pl_noise'), z=[t('z_pl')])
    worst['pl'] = _check_grads(tr.generator, s, 'pl', 'pl_grad_norm', 3 * tol)
    accumulate(tr.g_ema, tr.generator, tr.accum)
    print(name, 'worst per-parameter gradient-norm error per pass:', {k: '%.2e' % v for k, v in worst.items()})
    for k in ('d_loss', 'd_r1_loss', 'g_adv_loss', 'g_path_loss', 'g_mean_path_length'):
        ref = float(s[f'stat/{k}'])
        assert abs(float(tr.stats[k]) - ref) <= tol * max(1.0, abs(ref)), (k, float(tr.stats[k]), ref)
    assert rel_err(tr.stats['path_lengths'], t('stat/path_lengths')) <= tol
    # parameters after the four Adam updates (first Adam steps move every weight by ~lr, so an absolute bound)
    bad = 0
    total = 0
    for tag, mod in (('g', tr.generator), ('d', tr.discriminator), ('g_ema', tr.g_ema)):
        params = dict(mod.named_parameters())
        for name, val in zip(s[f'param/{tag}/names'], s[f'param/{tag}/vals']):
            key, idx = str(name).rsplit('#', 1)
            total += 1
            if abs(float(params[key].detach().reshape(-1)[int(idx)]) - float(val)) > 1e-3:
                bad += 1
    # sign flips of ~zero gradients under Adam's sign-like first steps may move a few samples by 2*lr
    assert bad <= total * 0.01, f'{bad} of {total} sampled parameters differ'
    return tr
