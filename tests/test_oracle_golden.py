"""The oracle (oracle/) must reproduce every golden vector captured from the reference."""
import numpy as np
import pytest
import torch
from torch import autograd

from conftest import load_golden, group, rel_err
from oracle import ops, networks, step


def names(gold):
    return sorted({k.split('/')[0] for k in gold})


def check_grads3(fn, inputs, rec, tol):
    inputs = [t.clone().requires_grad_(True) for t in inputs]
    out = fn(*inputs)
    assert rel_err(out, rec['out']) <= tol
    go = rec['go'].clone().requires_grad_(True)
    gi = autograd.grad(out, inputs, go, create_graph=True, allow_unused=True)
    for i, g in enumerate(gi):
        if f'gi{i}' in rec:
            assert rel_err(g, rec[f'gi{i}']) <= tol, f'gi{i}'
    gg = autograd.grad((gi[0] * rec['v']).sum(), [go] + inputs, allow_unused=True)
    for i, g in enumerate(gg):
        if f'gg{i}' in rec:
            assert g is not None, f'gg{i} is None'
            assert rel_err(g, rec[f'gg{i}']) <= tol, f'gg{i}'


UPF = load_golden('upfirdn2d')


@pytest.mark.parametrize('case', names(UPF))
def test_upfirdn2d(case):
    r = group(UPF, case)
    up, down, p0, p1 = [int(v) for v in r['args']]
    check_grads3(lambda t: ops.upfirdn2d(t, r['k'], up, down, (p0, p1)), [r['x']], r, 1e-6)


BA = load_golden('bias_act')


@pytest.mark.parametrize('case', names(BA))
def test_bias_act(case):
    r = group(BA, case)
    check_grads3(lambda t, b: ops.fused_leaky_relu(t, b), [r['x'], r['b']], r, 1e-6)


CV = load_golden('convs')


@pytest.mark.parametrize('case', [n for n in names(CV) if n.startswith('mod_')])
def test_modulated_conv(case):
    r = group(CV, case)
    demod, up, k = [int(v) for v in r['cfg']]
    fn = lambda x, st, w, mw, mb: ops.modulated_conv2d(x, st, w, mw, mb, demodulate=bool(demod), upsample=bool(up))
    check_grads3(fn, [r['x'], r['style'], r['weight'], r['mod_weight'], r['mod_bias']], r, 2e-5)


@pytest.mark.parametrize('case', [n for n in names(CV) if n.startswith('conv_')])
def test_equal_conv(case):
    r = group(CV, case)
    stride, padding, k = [int(v) for v in r['cfg']]
    ins = [r['x'], r['weight']] + ([r['bias']] if 'bias' in r else [])
    fn = lambda x, w, *b: ops.equal_conv2d(x, w, b[0] if b else None, stride=stride, padding=padding)
    check_grads3(fn, ins, r, 2e-5)


def test_misc():
    g = load_golden('misc')
    for name, act in (('lin_map', True), ('lin_plain', False)):
        r = group(g, name)
        assert rel_err(ops.equal_linear(r['x'], r['w'], r['b'], lr_mul=float(r['lr_mul']), activation=act), r['out']) < 1e-6
    r = group(g, 'pixel_norm')
    assert rel_err(ops.pixel_norm(r['x']), r['out']) < 1e-6
    for name in ('mbstd8', 'mbstd2', 'mbstd4'):
        r = group(g, name)
        assert rel_err(ops.minibatch_stddev(r['x']), r['out']) < 1e-6
    r = group(g, 'loss')
    assert rel_err(step.d_logistic_loss(r['real_pred'], r['fake_pred']), r['d_logistic']) < 1e-6
    assert rel_err(step.g_nonsaturating_loss(r['fake_pred']), r['g_nonsat']) < 1e-6
    pen, mean, lens = step.path_lengths_and_penalty(r['path_grad'], torch.tensor(0.37))  # fixture: warm running mean
    assert rel_err(pen, r['path_penalty']) < 1e-6 and rel_err(mean, r['path_mean']) < 1e-6 and rel_err(lens, r['path_lengths']) < 1e-6
    hp = step.adam_hparams(0.002, 16)
    assert abs(hp['lr'] - 0.002 * 16 / 17) < 1e-12 and hp['betas'] == (0.0, 0.99 ** (16 / 17))


def build_sd(size, fc_groups=None):
    """Procedurally filled state dicts with the reference's key names, built without any module."""
    from gan_control_amd.models.gan_model import Generator, Discriminator
    from gan_control_amd.utils.fc_config import FcConfig
    fc = None
    if fc_groups is not None:
        fc = FcConfig([n for n, _ in fc_groups], {n: {'latent_place': list(b), 'latent_size': b[1] - b[0]} for n, b in fc_groups})
    g = Generator(size, 512, 8, channel_multiplier=2, conv_transpose=True, split_fc=fc is not None, fc_config=fc)
    d = Discriminator(size, channel_multiplier=2)
    return networks.procedural_fill_(g.state_dict()), networks.procedural_fill_(d.state_dict())


def seeded_noise(size, batch, seed):
    import math
    gen = torch.Generator().manual_seed(seed)
    maps = [torch.randn(batch, 1, 4, 4, generator=gen)]
    for i in range(3, int(math.log2(size)) + 1):
        maps += [torch.randn(batch, 1, 2 ** i, 2 ** i, generator=gen) for _ in range(2)]
    return maps


NET = load_golden('networks')


@pytest.mark.parametrize('size', [32, 64])
def test_networks(size):
    r = group(NET, f's{size}')
    g_sd, d_sd = build_sd(size)
    noise = seeded_noise(size, int(r['batch']), int(r['noise_seed']))
    with torch.no_grad():
        img, lat = networks.generator_forward(g_sd, [r['z']], size, noise=noise)
        logits = networks.discriminator_forward(d_sd, img)
    assert rel_err(img, r['img']) < 1e-5
    assert rel_err(logits, r['logits']) < 1e-5
    assert rel_err(lat[:, 0, :16], r['w0']) < 1e-5
    assert abs(float(img.mean()) - float(r['img_mean'])) < 1e-4


def test_network_split_fc_and_mixing():
    r = group(NET, 'split32')
    fc_groups = [(str(n), tuple(int(v) for v in b)) for n, b in zip(NET['split32/group_names'], NET['split32/group_bounds'])]
    g_sd, _ = build_sd(32, fc_groups)
    with torch.no_grad():
        img, _ = networks.generator_forward(g_sd, [r['z']], 32, noise=seeded_noise(32, 2, int(r['noise_seed'])), fc_groups=fc_groups)
    assert rel_err(img, r['img']) < 1e-5
    r = group(NET, 'mix32')
    g_sd, _ = build_sd(32)
    with torch.no_grad():
        img, _ = networks.generator_forward(g_sd, [r['z'], r['z2']], 32, noise=seeded_noise(32, 2, int(r['noise_seed'])),
                                            inject_index=3, truncation=0.7, truncation_latent=r['mean_w'])
    assert rel_err(img, r['img']) < 1e-5


def test_noise_modes():
    gold = load_golden('noise_modes')
    for mode in ('zeros', 'id_zeros'):
        r = group(gold, mode)
        g_sd, _ = build_sd(32)
        with torch.no_grad():
            img, _ = networks.generator_forward(g_sd, [r['z']], 32, noise=seeded_noise(32, 2, int(r['noise_seed'])), noise_mode=mode)
        assert rel_err(img, r['img']) < 1e-5, mode


def test_step():
    s = load_golden('step')
    g_sd, d_sd = build_sd(32)
    o = step.OracleStep(g_sd, d_sd, 32, 4, none_g=[str(n) for n in s['none_g']], none_d=[str(n) for n in s['none_d']])
    t = lambda k: torch.from_numpy(s[k])
    n = [int(v) for v in s['noise_seeds']]
    o.iteration(0, t('real'), t('z_d'), t('z_g'), z_pl=t('z_pl'), noise_d=seeded_noise(32, 4, n[0]),
                noise_g=seeded_noise(32, 4, n[1]), noise_pl=seeded_noise(32, 2, n[2]), pl_noise=t('pl_noise'))
    for k in ('d_loss', 'd_r1_loss', 'g_adv_loss', 'g_path_loss', 'g_mean_path_length'):
        assert abs(o.stats[k] - float(s[f'stat/{k}'])) <= 2e-4 * max(1.0, abs(float(s[f'stat/{k}']))), k
    assert rel_err(o.stats['path_lengths'], t('stat/path_lengths')) < 2e-4
    for tag, params in (('g', o.g_params), ('d', o.d_params), ('g_ema', o.g_ema)):
        for name, val in zip(s[f'param/{tag}/names'], s[f'param/{tag}/vals']):
            key, idx = str(name).rsplit('#', 1)
            assert abs(float(params[key].detach().reshape(-1)[int(idx)]) - float(val)) < 5e-4, name
    # default none-grad name sets equal the ones the reference's dry_run finds
    o2 = step.OracleStep(g_sd, d_sd, 32, 4)
    assert sorted(o2.none_g) == sorted(str(n) for n in s['none_g'])
    assert sorted(o2.none_d) == sorted(str(n) for n in s['none_d'])
