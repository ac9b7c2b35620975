"""Step-level parity: the product trainer must reproduce the iteration captured from the reference."""
import os

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


class _Measure(dict):
    """check_step(measure=_Measure()): record every error next to its name instead of asserting (how the tolerances of a new arithmetic
    mode are derived: run once in measure mode, assert 2 x the measured values from then on)."""

    def note(self, key, err):
        self[key] = max(self.get(key, 0.0), float(err))


# The ratchet (round 6).  The asserted bounds of check_isolated are 2 x (3 x for the double-backward passes) what was once measured: a kernel change
# that doubled an error would still pass them.  So every run ALSO records each error it looks at (RECORD) and, given the committed table of
# measured values (tests/golden/parity_measured.json, written by tools/headline_parity_probe.py --write on the build that is committed with it),
# fails when one exceeds RATCHET x its committed value -- a conscious re-measurement is then the only way to move a bound.  Errors below RATCHET_FLOOR
# (loss scalars that agree to fp32 rounding) are not ratcheted: their value is noise.
RECORD = None
RATCHET, RATCHET_FLOOR = 1.3, 5e-6


def _record(key, err):
    if RECORD is not None:
        RECORD.note(key, err)


def load_measured():
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'parity_measured.json')) as f:
        return json.load(f)


def check_ratchet(label, seen, committed):
    """Print the table measured-now / committed / ratio; assert the ratchet."""
    rows, bad = [], []
    for key in sorted(committed):
        now, ref = seen.get(key), committed[key]
        if now is None:
            bad.append((key, 'not measured in this run'))
            continue
        bound = max(RATCHET * ref, RATCHET_FLOOR)
        rows.append('  %-22s now %.3e   committed %.3e   x%.2f%s' % (key, now, ref, now / ref if ref else float('inf'), '   <-- above the ratchet' if now > bound else ''))
        if now > bound:
            bad.append((key, now, ref))
    print('parity ratchet, %s (fails above %.1f x the committed value):\n%s' % (label, RATCHET, '\n'.join(rows)))
    assert not bad, ('measured errors above %.1f x tests/golden/parity_measured.json -- if the change is intended, re-measure with '
                     'tools/headline_parity_probe.py --write and commit the table' % RATCHET, label, bad)


ELEMS = 64


def elem_index(name, numel):
    """The sampled positions of parameter ``name`` (oracle/make_golden.py::elem_index: seeded by the name)."""
    import zlib
    gen = torch.Generator().manual_seed(zlib.crc32(('elem:' + name).encode()) & 0x7FFFFFFF)
    return torch.randint(0, numel, (ELEMS,), generator=gen)


def _check_elems(got, s, prefix, names, refs, ref_total, elem_tol, measure):
    """DIRECTION of every parameter's gradient, not just its length: 64 sampled elements per tensor (``<prefix>/elems``) against the
    reference's, as the relative L2 distance of the two sample vectors.  A permuted, transposed or sign-flipped block with the right norm
    passes the norm check; it cannot pass this one.  Tensors whose whole gradient is below 0.1 % of the pass's norm are bounded against
    that floor (their samples sit in the rounding noise of the pass), the scalar noise strengths are compared together as one vector."""
    if f'{prefix}/elems' not in s:
        return 0.0
    elems = torch.from_numpy(s[f'{prefix}/elems']).double()
    worst, scal_got, scal_ref = 0.0, [], []
    for i, n in enumerate(names):
        g = got[n].detach().reshape(-1)
        ref = elems[i]
        mine = g[elem_index(n, g.numel()).to(g.device)].double().cpu()
        if g.numel() == 1:
            scal_got.append(mine[0]); scal_ref.append(ref[0])
            continue
        # the expected length of a 64-sample vector of a tensor at the floor norm
        floor = 1e-3 * ref_total * (ELEMS / g.numel()) ** 0.5
        err = float((mine - ref).norm() / max(float(ref.norm()), floor))
        worst = max(worst, err)
        assert measure is not None or err <= elem_tol, (prefix, n, 'sampled gradient elements', err, elem_tol)
    if scal_ref:
        a, b = torch.stack(scal_got), torch.stack(scal_ref)
        err = float((a - b).norm() / b.norm().clamp_min(1e-3 * ref_total))
        worst = max(worst, err)
        assert measure is not None or err <= elem_tol, (prefix, 'scalar parameters, elementwise', err)
    if measure is not None:
        measure.note(prefix + ':elems', worst)
    _record(prefix + ':elems', worst)
    return worst


def _check_grads(module, s, prefix, ref_total, tol, param_tol=None, measure=None, elem_tol=None):
    """Gradients of one backward pass against the reference's: the set of parameters that have one, the global norm, the
    norm of every single parameter's gradient (``<prefix>/names``, ``<prefix>/vals``) and 64 sampled elements of each (``<prefix>/elems``)."""
    got = {n: p.grad for n, p in module.named_parameters() if p.grad is not None}
    names = [str(n) for n in s[f'{prefix}/names']]
    assert sorted(names) == sorted(got), (prefix, set(names) ^ set(got))
    refs = [float(v) for v in s[f'{prefix}/vals']]
    if ref_total is None:
        ref_total = float(torch.tensor(refs, dtype=torch.float64).norm())
    total = float(torch.stack([g.double().pow(2).sum() for g in got.values()]).sum().sqrt())
    _record(prefix + ':global', abs(total - ref_total) / ref_total)
    if measure is not None:
        measure.note(prefix + ':global', abs(total - ref_total) / ref_total)
    else:
        assert abs(total - ref_total) <= tol * ref_total, (prefix, total, ref_total)
    # A parameter whose gradient is below 0.1 % of the pass's norm is bounded absolutely: such gradients (e.g. what reaches D's
    # activation biases under R1 only through the minibatch-stddev channel, ~1e-4 of the total) sit below the rounding noise of the pass.
    floor = 1e-3 * ref_total
    param_tol = tol if param_tol is None else param_tol
    worst = 0.0
    scalars = []
    for n, ref in zip(names, refs):
        val = float(got[n].double().norm())
        if got[n].numel() == 1:
            # a NoiseInjection strength: ONE scalar = a sum over batch x channels x pixels of random-sign terms, cancelling to a small
            # fraction of their magnitude: the 13-17 strengths of the network are compared together as one vector
            scalars.append((val, ref))
            continue
        err = abs(val - ref) / max(ref, floor)
        worst = max(worst, err)
        assert measure is not None or err <= param_tol, (prefix, n, val, ref)
    if scalars:
        a, b = torch.tensor(scalars, dtype=torch.float64).unbind(1)
        err = float((a - b).norm() / b.norm().clamp_min(floor))
        worst = max(worst, err)
        assert measure is not None or err <= param_tol, (prefix, 'scalar parameters', scalars)
    if measure is not None:
        measure.note(prefix + ':param', worst)
    _record(prefix + ':param', worst)
    # sampled elements: a sample vector's relative error is bounded like a per-parameter norm's (the same cancellation argument), with the
    # sqrt(2) of a difference of two roundings
    _check_elems(got, s, prefix, names, refs, ref_total, (2 * param_tol) if elem_tol is None else elem_tol, measure)
    return worst


def check_isolated(device, name='step_1024_b4', tol=2e-3, param_tol=None, measure=None, trainer=None, ratchet=None):
    """The four backward passes of an iteration, EACH IN ISOLATION from the procedural weights, against ``tests/golden/<name>.npz``
    (oracle/make_golden.py::golden_step_isolated: the reference's own modules and trainer maths at the workload bench.py times -- 1024 x 1024,
    4 images).  ``trainer``: a factory ``(size, batch) -> GeneratorTrainer`` -- the headline test passes bench.py's own construction (fused Adam,
    every default knob); the weights are then overwritten with the procedural ones the fixture was generated from.  Checked per pass: loss /
    prediction statistics, the set of parameters with a gradient, global and per-parameter gradient norms, 64 sampled gradient elements per
    parameter tensor.  The Adam steps the trainer takes after each backward are undone by reloading the weights (gradients are read first)."""
    from gan_control_amd.trainers.utils import requires_grad
    from oracle.networks import procedural_fill_
    pt = param_tol if param_tol is not None else tol
    m = measure
    s = load_golden(name)
    size, batch = [int(v) for v in s['cfg']]
    gen = torch.Generator().manual_seed(int(s['input_seed']))
    real = torch.rand(batch, 3, size, size, generator=gen) * 2 - 1
    z_d, z_g = torch.randn(batch, 512, generator=gen), torch.randn(batch, 512, generator=gen)
    z_pl = torch.randn(batch // 2, 512, generator=gen)
    pl_noise = torch.randn(batch // 2, 3, size, size, generator=gen)
    for k, v in (('z_d', z_d), ('z_g', z_g), ('z_pl', z_pl)):
        assert torch.equal(v, torch.from_numpy(s[k])), 'the seeded redraw does not reproduce the fixture: ' + k
    tr = trainer(size, batch) if trainer is not None else make_trainer(device, size=size, batch=batch)
    fresh_g = procedural_fill_({k: v.detach().clone() for k, v in tr.generator.state_dict().items()})
    fresh_d = procedural_fill_({k: v.detach().clone() for k, v in tr.discriminator.state_dict().items()})
    assert sorted(tr.none_g_grads) == sorted(str(n) for n in s['none_g'])
    assert sorted(tr.none_d_grads) == sorted(str(n) for n in s['none_d'])
    seeds = [int(v) for v in s['noise_seeds']]
    noise = lambda b, i: oc.seeded_noise(size, b, seeds[i], device)
    real, z_d, z_g, z_pl, pl_noise = (t.to(device) for t in (real, z_d, z_g, z_pl, pl_noise))

    global RECORD
    seen = RECORD = _Measure()

    def within(key, err, bound):
        _record(key, err)
        if m is not None:
            m.note(key, err)
        else:
            assert err <= bound, (key, err, bound)

    def reset():
        tr.generator.load_state_dict(fresh_g)
        tr.discriminator.load_state_dict(fresh_d)
        tr.g_ema.load_state_dict(fresh_g)
        tr.mean_path_length = 0

    def stat(key, scale=1.0):
        ref = torch.from_numpy(s['iso/stat/' + key]).double().reshape(-1)
        return ref, max(scale, float(ref.abs().max()))

    worst = {}
    reset()
    requires_grad(tr.generator, False); requires_grad(tr.discriminator, True)
    tr.discriminator_step([[z_d]], [real], noise=noise(batch, 0))
    worst['d'] = _check_grads(tr.discriminator, s, 'iso/d', None, tol, pt, m)
    ref, sc = stat('d_loss')
    within('iso d_loss', abs(float(tr.stats['d_loss']) - float(ref)) / sc, tol)
    reset()
    tr.discriminator_regularize_step([real])
    worst['r1'] = _check_grads(tr.discriminator, s, 'iso/r1', None, 3 * tol, 3 * pt, m)
    ref, _ = stat('d_r1_loss')
    within('iso d_r1_loss', abs(float(tr.stats['d_r1_loss']) - float(ref)) / max(1e-3, abs(float(ref))), 3 * tol)
    reset()
    requires_grad(tr.generator, True); requires_grad(tr.discriminator, False)
    tr.generator_step([[z_g]], noise=noise(batch, 1))
    worst['g'] = _check_grads(tr.generator, s, 'iso/g', None, tol, pt, m)
    ref, sc = stat('g_adv_loss')
    within('iso g_adv_loss', abs(float(tr.stats['g_adv_loss']) - float(ref)) / sc, tol)
    reset()
    tr.generator_regularize_step(noise=noise(batch // 2, 2), pl_noise=pl_noise, z=[z_pl])
    worst['pl'] = _check_grads(tr.generator, s, 'iso/pl', None, 3 * tol, 3 * pt, m)
    ref, sc = stat('g_path_loss')
    within('iso g_path_loss', abs(float(tr.stats['g_path_loss']) - float(ref)) / sc, tol)
    within('iso path_lengths', rel_err(tr.stats['path_lengths'], torch.from_numpy(s['iso/stat/path_lengths'])), tol)
    print(name, 'worst per-parameter gradient-norm error per pass:', {k: '%.2e' % v for k, v in worst.items()})
    RECORD = None
    if ratchet is not None:
        check_ratchet(name, seen, ratchet)
    return tr


def check_step(device, tol=2e-3, name='step', param_tol=None, measure=None):
    """One full iteration (D step, R1, G step, path-length, EMA) of the product trainer against the iteration captured from
    the reference.  ``name`` selects the fixture: 'step' (32x32, batch 4), 'step_512' (512x512, batch 4), 'step_1024'
    (1024x1024, batch 2).

    1. The SEQUENTIAL iteration: loss scalars, path lengths, the None-gradient name sets, sampled parameter values after the four
       Adam updates, and the gradients of the first backward pass (D step; global and per-parameter norms).
    2. Every other backward pass IN ISOLATION (R1, G step, path length from the un-updated procedural weights): losses and
       global / per-parameter gradient norms.  The first Adam steps are sign-like, so inside the sequential iteration an element
       whose gradient is ~0 may step the other way and every later pass inherits that noise (which is why step 1 compares
       parameters with a budget of outliers); the isolated passes pin each backward without it.
    Tolerances: `tol` for the plain backward passes, 3 * tol for the two double-backward passes: in fp64 the product reproduces the
    oracle's double-backward gradients to 1e-15, in fp32 ONE leaky-ReLU whose pre-activation is within rounding of zero (expected
    ~0.4 per million activations) takes the other slope and moves a whole layer's second-order gradient by ~5e-4
    (tools/pl_error_probe.py).  ``param_tol`` (default: tol) bounds the per-parameter gradient norms separately from the losses and
    the global norms: a single parameter's gradient is a sum over batch x pixels that cancels to ~1/700 of its terms on the worst
    parameter, which amplifies the arithmetic's rounding error (fp32: 3e-7 -> 2e-4 measured; split-bf16: 5e-6 -> 4e-3 measured)."""
    pt = param_tol if param_tol is not None else tol
    m = measure

    def within(key, err, bound, detail):
        if m is not None:
            m.note(key, err / 1.0)
            m.setdefault('_bounds', {})[key] = bound
        else:
            assert err <= bound, (key, err, bound, detail)

    from gan_control_amd.trainers.utils import requires_grad, accumulate
    s = load_golden(name)
    size, batch = [int(v) for v in s['cfg']]
    inputs = step_inputs(s, size, batch)
    t = lambda k: (inputs[k] if k in inputs else torch.from_numpy(s[k])).to(device)
    tr = make_trainer(device, size=size, batch=batch)
    fresh_g = {k: v.detach().clone() for k, v in tr.generator.state_dict().items()}
    fresh_d = {k: v.detach().clone() for k, v in tr.discriminator.state_dict().items()}
    assert sorted(tr.none_g_grads) == sorted(str(n) for n in s['none_g'])
    assert sorted(tr.none_d_grads) == sorted(str(n) for n in s['none_d'])
    seeds = [int(v) for v in s['noise_seeds']]
    real = t('real')
    noise = lambda b, i: oc.seeded_noise(size, b, seeds[i], device)
    # 1. iteration 0 with the fixture's latents and noise maps
    worst = {}
    requires_grad(tr.generator, False); requires_grad(tr.discriminator, True)
    tr.discriminator_step([[t('z_d')]], [real], noise=noise(batch, 0))
    worst['d'] = _check_grads(tr.discriminator, s, 'gradnorm/d', float(s['stat/d_grad_norm']), tol, pt, m)
    tr.discriminator_regularize_step([real])
    requires_grad(tr.generator, True); requires_grad(tr.discriminator, False)
    tr.generator_step([[t('z_g')]], noise=noise(batch, 1))
    tr.generator_regularize_step(noise=noise(batch // 2, 2), pl_noise=t('pl_noise'), z=[t('z_pl')])
    accumulate(tr.g_ema, tr.generator, tr.accum)
    seq = {k: (tr.stats[k].clone() if torch.is_tensor(tr.stats[k]) else tr.stats[k]) for k in tr.stats}
    after = {tag: {k: v.detach().clone() for k, v in mod.named_parameters()} for tag, mod in (('g', tr.generator), ('d', tr.discriminator), ('g_ema', tr.g_ema))}

    # 2. the other three passes in isolation
    def reset():
        tr.generator.load_state_dict(fresh_g)
        tr.discriminator.load_state_dict(fresh_d)
        tr.mean_path_length = 0

    reset()
    requires_grad(tr.generator, False); requires_grad(tr.discriminator, True)
    tr.discriminator_regularize_step([real])
    worst['r1'] = _check_grads(tr.discriminator, s, 'iso/r1', None, 3 * tol, 3 * pt, m)
    ref = float(s['iso/stat/d_r1_loss'])
    within('iso d_r1_loss', abs(float(tr.stats['d_r1_loss']) - ref) / max(1e-3, abs(ref)), tol, ref)
    reset()
    requires_grad(tr.generator, True); requires_grad(tr.discriminator, False)
    tr.generator_step([[t('z_g')]], noise=noise(batch, 1))
    worst['g'] = _check_grads(tr.generator, s, 'iso/g', None, tol, pt, m)
    ref = float(s['iso/stat/g_adv_loss'])
    within('iso g_adv_loss', abs(float(tr.stats['g_adv_loss']) - ref) / max(1.0, abs(ref)), tol, ref)
    reset()
    tr.generator_regularize_step(noise=noise(batch // 2, 2), pl_noise=t('pl_noise'), z=[t('z_pl')])
    worst['pl'] = _check_grads(tr.generator, s, 'iso/pl', None, 3 * tol, 3 * pt, m)
    ref = float(s['iso/stat/g_path_loss'])
    within('iso g_path_loss', abs(float(tr.stats['g_path_loss']) - ref) / max(1.0, abs(ref)), tol, ref)
    within('iso path_lengths', rel_err(tr.stats['path_lengths'], torch.from_numpy(s['iso/stat/path_lengths'])), tol, None)
    print(name, 'worst per-parameter gradient-norm error per pass:', {k: '%.2e' % v for k, v in worst.items()})
    tr.stats = seq
    # the path-length pass of the SEQUENTIAL iteration runs on a generator that has just taken a sign-like Adam step (every one of its
    # 30 M weights moved by +-lr; near-zero gradients may step either way): its statistics get 3 * tol here -- the isolated pass above,
    # free of that noise, is held to tol
    for k, f in (('d_loss', 1), ('d_r1_loss', 1), ('g_adv_loss', 1), ('g_path_loss', 3), ('g_mean_path_length', 3)):
        ref = float(s[f'stat/{k}'])
        within('seq ' + k, abs(float(tr.stats[k]) - ref) / max(1.0, abs(ref)), f * tol, ref)
    within('seq path_lengths', rel_err(tr.stats['path_lengths'], t('stat/path_lengths')), 3 * tol, None)
    # Parameters after the four Adam updates (first Adam steps move every weight by ~lr, so an absolute bound).  Adam's first steps are
    # sign-like: a sampled element can only land elsewhere (by 2 * lr) if one of the gradients that stepped it is within the arithmetic's
    # error of zero.  The fixture holds, per sampled element, min over its passes of |gradient element| / RMS(gradient tensor)
    # (`param/<net>/grad_ratio`, oracle/make_golden.py::golden_step): the budget of outliers is the NUMBER OF SAMPLES whose ratio is below
    # the per-parameter gradient tolerance of this mode (3 * param_tol: the double-backward passes' bound) -- not a flat percentage.
    bad, total, budget, offenders = 0, 0, 0, []
    for tag in ('g', 'd', 'g_ema'):
        params = after[tag]
        ratios = s[f'param/{tag}/grad_ratio']
        for name, val, ratio in zip(s[f'param/{tag}/names'], s[f'param/{tag}/vals'], ratios):
            key, idx = str(name).rsplit('#', 1)
            total += 1
            near_zero = float(ratio) < 3 * pt
            budget += near_zero
            if abs(float(params[key].detach().reshape(-1)[int(idx)]) - float(val)) > 1e-3:
                bad += 1
                if not near_zero:
                    offenders.append((tag, str(name), float(ratio)))
    if m is not None:
        m['param outliers'] = bad
        m['param outliers with a gradient above the noise'] = len(offenders)
        m['param budget (samples whose gradient is within 3 * param_tol of zero)'] = budget
        m['param samples'] = total
    else:
        assert not offenders, f'sampled parameters differ although their gradients are well above the arithmetic\'s error: {offenders[:5]}'
        assert bad <= budget, f'{bad} of {total} sampled parameters differ; only {budget} have a gradient near zero'
    return tr
