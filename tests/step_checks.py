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


def check_step(device, tol=2e-3):
    from gan_control_amd.trainers.utils import requires_grad, accumulate
    s = load_golden('step')
    t = lambda k: torch.from_numpy(s[k]).to(device)
    tr = make_trainer(device)
    assert sorted(tr.none_g_grads) == sorted(str(n) for n in s['none_g'])
    assert sorted(tr.none_d_grads) == sorted(str(n) for n in s['none_d'])
    seeds = [int(v) for v in s['noise_seeds']]
    real = t('real')
    # iteration 0 with the fixture's latents and noise maps
    requires_grad(tr.generator, False); requires_grad(tr.discriminator, True)
    tr.discriminator_step([[t('z_d')]], [real], noise=oc.seeded_noise(32, 4, seeds[0], device))
    tr.discriminator_regularize_step([real])
    requires_grad(tr.generator, True); requires_grad(tr.discriminator, False)
    tr.generator_step([[t('z_g')]], noise=oc.seeded_noise(32, 4, seeds[1], device))
    tr.generator_regularize_step(noise=oc.seeded_noise(32, 2, seeds[2], device), pl_noise=t('pl_noise'), z=[t('z_pl')])
    accumulate(tr.g_ema, tr.generator, tr.accum)
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
