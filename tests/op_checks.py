"""Parity checks shared by the CPU (emulated C ABI) and GPU (HIP kernels) test modules.

Every check feeds golden inputs captured from the reference through the PRODUCT's operator
socket (gan_control_amd.models.op) and compares forward, first-order and second-order
gradients with the golden outputs.
"""
import math

import numpy as np
import torch
from torch import autograd

from conftest import load_golden, group, rel_err

TOL = 1e-3          # north-star bound: 1e-3 relative fp32; the kernels are ~1e-6
TIGHT = 2e-5


def names(gold, prefix=''):
    return sorted({k.split('/')[0] for k in gold if k.startswith(prefix)})


def run_grads3(fn, inputs, rec, device, tol):
    inputs = [t.clone().to(device).requires_grad_(True) for t in inputs]
    out = fn(*inputs)
    assert out.shape == rec['out'].shape
    assert rel_err(out, rec['out']) <= tol, 'forward'
    go = rec['go'].clone().to(device).requires_grad_(True)
    gi = autograd.grad(out, inputs, go, create_graph=True, allow_unused=True)
    for i, g in enumerate(gi):
        if f'gi{i}' in rec:
            assert g is not None, f'gi{i} is None'
            assert rel_err(g, rec[f'gi{i}']) <= tol, f'gi{i}'
    gg = autograd.grad((gi[0] * rec['v'].to(device)).sum(), [go] + inputs, allow_unused=True)
    for i, g in enumerate(gg):
        if f'gg{i}' in rec:
            assert g is not None, f'gg{i} is None'
            assert rel_err(g, rec[f'gg{i}']) <= tol, f'gg{i}'


UPF = load_golden('upfirdn2d')
BA = load_golden('bias_act')
CV = load_golden('convs')
NET = load_golden('networks')


def check_upfirdn2d(case, device):
    from gan_control_amd.models.op import upfirdn2d
    r = group(UPF, case)
    up, down, p0, p1 = [int(v) for v in r['args']]
    k = r['k'].to(device)
    run_grads3(lambda t: upfirdn2d(t, k, up=up, down=down, pad=(p0, p1)), [r['x']], r, device, TIGHT)


def check_bias_act(case, device):
    from gan_control_amd.models.op import fused_leaky_relu, FusedLeakyReLU
    r = group(BA, case)
    run_grads3(lambda t, b: fused_leaky_relu(t, b), [r['x'], r['b']], r, device, TIGHT)
    if r['x'].ndim == 4:
        m = FusedLeakyReLU(r['x'].shape[1]).to(device)
        with torch.no_grad():
            m.bias.copy_(r['b'])
        assert rel_err(m(r['x'].to(device)), r['out']) <= TIGHT


def check_noise_bias_act(device):
    """fused noise + bias + act == FusedLeakyReLU(NoiseInjection(x)) (gan_model.py:340-345,402-408), all gradients."""
    from gan_control_amd.models.op import fused_noise_bias_act
    from oracle import ops
    gen = torch.Generator().manual_seed(3)
    for shape in [(2, 5, 6, 7), (2, 3, 32, 40), (1, 4, 4, 4)]:
        x = torch.randn(*shape, generator=gen)
        b = torch.randn(shape[1], generator=gen)
        nz = torch.randn(shape[0], 1, shape[2], shape[3], generator=gen)
        nw = torch.randn(1, generator=gen)
        ins_o = [t.clone().double().requires_grad_(True) for t in (x, b, nw)]
        out_o = ops.fused_leaky_relu(ins_o[0] + ins_o[2] * nz.double(), ins_o[1])
        ins_p = [t.clone().to(device).requires_grad_(True) for t in (x, b, nw)]
        out_p = fused_noise_bias_act(ins_p[0], ins_p[1], nz.to(device), ins_p[2])
        assert rel_err(out_p, out_o) <= TIGHT
        go = torch.randn(*shape, generator=gen)
        g_o = autograd.grad(out_o, ins_o, go.double(), create_graph=True)
        go_p = go.to(device).requires_grad_(True)
        g_p = autograd.grad(out_p, ins_p, go_p, create_graph=True)
        for a, c in zip(g_p, g_o):
            assert rel_err(a, c) <= TIGHT
        # second order: d/d(go) of <gx, v>
        v = torch.randn(*shape, generator=gen)
        gg_p, = autograd.grad((g_p[0] * v.to(device)).sum(), [go_p])
        mask = torch.where(out_o > 0, math.sqrt(2), 0.2 * math.sqrt(2))
        assert rel_err(gg_p, v.double() * mask) <= TIGHT


def check_equal_conv(case, device):
    from gan_control_amd.models.gan_model import EqualConv2d
    r = group(CV, case)
    stride, padding, k = [int(v) for v in r['cfg']]
    oc, ic = r['weight'].shape[:2]
    m = EqualConv2d(ic, oc, k, stride=stride, padding=padding, bias='bias' in r).to(device)

    def fn(x, w, *b):
        p = {'weight': w}
        if b:
            p['bias'] = b[0]
        return torch.func.functional_call(m, p, (x,))

    run_grads3(fn, [r['x'], r['weight']] + ([r['bias']] if 'bias' in r else []), r, device, TIGHT)


def check_modulated_conv(case, device):
    from gan_control_amd.models.gan_model import ModulatedConv2d
    r = group(CV, case)
    demod, up, k = [int(v) for v in r['cfg']]
    _, oc, ic = r['weight'].shape[:3]
    m = ModulatedConv2d(ic, oc, k, r['style'].shape[1], demodulate=bool(demod), upsample=bool(up), conv_transpose=True).to(device)

    def fn(x, st, w, mw, mb):
        return torch.func.functional_call(m, {'weight': w, 'modulation.weight': mw, 'modulation.bias': mb}, (x, st))

    ins = [r['x'], r['style'], r['weight'], r['mod_weight'], r['mod_bias']]
    run_grads3(fn, ins, r, device, TIGHT)
    # the no-grad route uses the fully fused kernel arguments (in_scale / out_scale): same numbers
    with torch.no_grad():
        out = fn(*[t.to(device) for t in ins])
    assert rel_err(out, r['out']) <= TIGHT


def conv_functional_checks(device):
    """conv2d_gradfix with the torch.nn.functional call signatures vs ATen on the CPU (fp64)."""
    import torch.nn.functional as F
    from gan_control_amd.models.op import conv2d_gradfix
    gen = torch.Generator().manual_seed(17)
    for (b, ic, oc, k, h, w, s, p) in [(2, 5, 4, 3, 9, 12, 1, 1), (1, 3, 7, 3, 10, 8, 2, 0), (2, 4, 3, 1, 7, 7, 1, 0),
                                       (1, 70, 33, 3, 6, 37, 1, 1), (3, 2, 2, 3, 5, 5, 2, 1)]:
        x = torch.randn(b, ic, h, w, generator=gen)
        wt = torch.randn(oc, ic, k, k, generator=gen)
        bias = torch.randn(oc, generator=gen)
        ref = F.conv2d(x.double(), wt.double(), bias.double(), stride=s, padding=p)
        out = conv2d_gradfix.conv2d(x.to(device), wt.to(device), bias.to(device), stride=s, padding=p)
        assert rel_err(out, ref) <= TIGHT, ('conv2d', b, ic, oc, k, h, w, s, p)
        wtt = torch.randn(ic, oc, k, k, generator=gen)
        ref = F.conv_transpose2d(x.double(), wtt.double(), bias.double(), stride=s, padding=p)
        out = conv2d_gradfix.conv_transpose2d(x.to(device), wtt.to(device), bias.to(device), stride=s, padding=p)
        assert rel_err(out, ref) <= TIGHT, ('conv_transpose2d', b, ic, oc, k, h, w, s, p)
        # gradients of the transposed conv (weight-grad goes through the operand-swapped wgrad)
        xr, wr = x.double().requires_grad_(True), wtt.double().requires_grad_(True)
        xp, wp = x.to(device).requires_grad_(True), wtt.to(device).requires_grad_(True)
        go = torch.randn_like(ref)
        gr = autograd.grad(F.conv_transpose2d(xr, wr, stride=s, padding=p), [xr, wr], go)
        gp = autograd.grad(conv2d_gradfix.conv_transpose2d(xp, wp, stride=s, padding=p), [xp, wp], go.float().to(device))
        for a, c in zip(gp, gr):
            assert rel_err(a, c) <= TIGHT, ('conv_transpose2d grad', b, ic, oc, k, h, w, s, p)


def seeded_noise(size, batch, seed, device='cpu'):
    gen = torch.Generator().manual_seed(seed)
    maps = [torch.randn(batch, 1, 4, 4, generator=gen)]
    for i in range(3, int(math.log2(size)) + 1):
        maps += [torch.randn(batch, 1, 2 ** i, 2 ** i, generator=gen) for _ in range(2)]
    return [m.to(device) for m in maps]


def check_noise_modes(device, tol=TOL):
    """StyledConv noise_mode 'zeros' / 'id_zeros' (gan_model.py:391-399, ModulatedNoiseInjection :1019-1035) against the reference generator
    built in that mode: image, the parameters left without a gradient, the gradient norms of a few named parameters, every noise strength's
    gradient (only conv1 and the up-sampling layers take the mode; the second convolution of a block is always 'normal', gan_model.py:606-610)."""
    from gan_control_amd.models.gan_model import Generator
    from oracle.networks import procedural_fill_
    gold = load_golden('noise_modes')
    for mode in ('zeros', 'id_zeros'):
        r = group(gold, mode)
        g = Generator(32, 512, 8, channel_multiplier=2, conv_transpose=True, noise_mode=mode)
        g.load_state_dict(procedural_fill_(g.state_dict()))
        g = g.to(device)
        img, _ = g([r['z'].to(device)], noise=seeded_noise(32, 2, int(r['noise_seed']), device))
        assert rel_err(img, r['img']) <= tol, mode
        (img * r['probe'].to(device)).sum().backward()
        named = dict(g.named_parameters())
        assert sorted(n for n, p in named.items() if p.grad is None) == [str(n) for n in gold[f'{mode}/none_grad']], mode
        for n, want in zip(gold[f'{mode}/grad_names'], r['grad_norms']):
            assert abs(float(named[str(n)].grad.norm()) - float(want)) <= 2 * tol * float(want), (mode, n)
        got = [float(p.grad) if p.grad is not None else float('nan') for n, p in named.items() if n.endswith('noise.weight')]
        want = r['noise_grads'].double().numpy()
        assert len(got) == len(want)
        scale = np.nanmax(np.abs(want))
        for a, b in zip(got, want):
            assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 2 * tol * scale, (mode, got, want)


def build_models(size, device, fc_groups=None):
    from gan_control_amd.models.gan_model import Generator, Discriminator
    from gan_control_amd.utils.fc_config import FcConfig
    from oracle.networks import procedural_fill_
    fc = None
    if fc_groups is not None:
        fc = FcConfig([n for n, _ in fc_groups], {n: {'latent_place': list(b), 'latent_size': b[1] - b[0]} for n, b in fc_groups})
    g = Generator(size, 512, 8, channel_multiplier=2, conv_transpose=True, split_fc=fc is not None, fc_config=fc)
    d = Discriminator(size, channel_multiplier=2)
    g.load_state_dict(procedural_fill_(g.state_dict()))
    d.load_state_dict(procedural_fill_(d.state_dict()))
    return g.to(device), d.to(device)


def network_errors(size, device):
    """The quantities check_network asserts, returned instead (used to derive the tolerance of the plain-bf16 mode from measurement)."""
    import torch.nn.functional as F
    r = group(NET, f's{size}')
    g, d = build_models(size, device)
    batch = int(r['batch'])
    noise = seeded_noise(size, batch, int(r['noise_seed']), device)
    with torch.no_grad():
        img, lat = g([r['z'].to(device)], noise=noise, return_latents=True)
        logits, _ = d(img)
    amax = float(r['img_absmax'])
    px = img.reshape(-1)[r['px_idx'].to(device)]
    return {'thumb': rel_err(F.adaptive_avg_pool2d(img, min(16, size)), r['thumb']), 'pixels': (px.cpu() - r['px_val']).abs().max().item() / amax,
            'mean': abs(float(img.mean()) - float(r['img_mean'])) / amax, 'std': abs(float(img.std()) - float(r['img_std'])) / amax,
            'logits': rel_err(logits, r['logits']), 'w0': rel_err(lat[:, 0, :16], r['w0'])}


def check_network(size, device, tol=TOL):
    r = group(NET, f's{size}')
    g, d = build_models(size, device)
    batch = int(r['batch'])
    noise = seeded_noise(size, batch, int(r['noise_seed']), device)
    with torch.no_grad():
        img, lat = g([r['z'].to(device)], noise=noise, return_latents=True)
        logits, _ = d(img)
    assert img.shape == (batch, 3, size, size)
    if 'img' in r:
        assert rel_err(img, r['img']) <= tol
    assert abs(float(img.mean()) - float(r['img_mean'])) <= tol * float(r['img_absmax'])
    assert abs(float(img.std()) - float(r['img_std'])) <= tol * float(r['img_absmax'])
    import torch.nn.functional as F
    assert rel_err(F.adaptive_avg_pool2d(img, min(16, size)), r['thumb']) <= tol
    px = img.reshape(-1)[r['px_idx'].to(device)]
    assert (px.cpu() - r['px_val']).abs().max().item() <= tol * float(r['img_absmax'])
    assert rel_err(logits, r['logits']) <= tol
    assert rel_err(lat[:, 0, :16], r['w0']) <= tol
    return g, d, img


MISC = load_golden('misc')


def fixture_fc_groups():
    return [(str(n), tuple(int(v) for v in b)) for n, b in zip(NET['split32/group_names'], NET['split32/group_bounds'])]


def check_split_fc(device, tol=TOL):
    """Product Generator with the per-group mapping networks (split_fc, MultiFcStack gan_model.py:489-502, 619-631) against the
    reference image; the groups are built by the product's own config ingestion from a sub_groups_dict."""
    from gan_control_amd.utils.fc_config import fc_config_from_sub_groups
    r = group(NET, 'split32')
    groups = fixture_fc_groups()
    # the sub_groups_dict the fixture was generated from, listed out of latent order on purpose
    sub = {n: {'place_in_latent': list(b), 'place_in_mini_batch': None} for n, b in reversed(groups)}
    fc = fc_config_from_sub_groups(sub, 512)
    assert fc.in_order_group_names == [n for n, _ in groups]
    assert [tuple(fc.groups[n]['latent_place']) for n in fc.in_order_group_names] == [b for _, b in groups]
    g, _ = build_models(32, device, fc_groups=groups)
    with torch.no_grad():
        img, _ = g([r['z'].to(device)], noise=seeded_noise(32, 2, int(r['noise_seed']), device))
    assert rel_err(img, r['img']) <= tol
    return g


def check_mixing_truncation(device, tol=TOL):
    """Two-style forward with inject_index and truncation towards a mean latent (gan_model.py:744-769)."""
    r = group(NET, 'mix32')
    g, _ = build_models(32, device)
    noise = seeded_noise(32, 2, int(r['noise_seed']), device)
    with torch.no_grad():
        img, lat = g([r['z'].to(device), r['z2'].to(device)], noise=noise, inject_index=3, truncation=0.7,
                     truncation_latent=r['mean_w'].to(device), return_latents=True)
        assert rel_err(img, r['img']) <= tol
        assert lat.shape == (2, g.n_latent, 512) and torch.equal(lat[:, 0], lat[:, 2]) and torch.equal(lat[:, 3], lat[:, -1])
        # the same latent fed back with input_is_latent reproduces the image (gan_model.py:738-743, 757-760)
        img2, _ = g([lat], input_is_latent=True, noise=noise)
        assert rel_err(img2, r['img']) <= tol
        # randomize_noise=False uses the registered [1, 1, h, w] noise buffers, broadcast over the batch (gan_model.py:728-734)
        stored = [getattr(g.noises, f'noise_{i}') for i in range(g.num_layers)]
        a, _ = g([r['z'].to(device)], randomize_noise=False)
        b, _ = g([r['z'].to(device)], noise=[n.expand(2, -1, -1, -1) for n in stored])
        assert rel_err(a, b) <= 1e-6


def check_transfer_learning(device, tol=TOL):
    """load_transfer_learning_model (gan_model.py:645-656): a regular-mapping checkpoint into a split_fc generator keeps the
    synthesis network and leaves the new mapping network alone; a mismatch in the main network is refused."""
    from gan_control_amd.models.gan_model import Generator
    src, _ = build_models(32, device)
    dst, _ = build_models(32, device, fc_groups=fixture_fc_groups())
    before = {k: v.clone() for k, v in dst.state_dict().items() if k.startswith('style.')}
    with torch.no_grad():
        for p in dst.convs.parameters():
            p.add_(1.0)
    dst.load_transfer_learning_model(src)
    sd_src, sd_dst = src.state_dict(), dst.state_dict()
    for k, v in sd_dst.items():
        if k.startswith('style.'):
            assert torch.equal(v, before[k]), k
        else:
            assert torch.equal(v, sd_src[k]), k
    # both now synthesise the same image from the same w
    r = group(NET, 's32')
    noise = seeded_noise(32, int(r['batch']), int(r['noise_seed']), device)
    with torch.no_grad():
        w = src.style(r['z'].to(device))
        a, _ = src([w], input_is_latent=True, noise=noise)
        b, _ = dst([w], input_is_latent=True, noise=noise)
    assert rel_err(a, r['img']) <= tol and torch.equal(a, b)
    other = Generator(64, 512, 8, channel_multiplier=2, conv_transpose=True).to(device)
    try:
        dst.load_transfer_learning_model(other)
    except ValueError as e:
        assert 'main network' in str(e)
    else:
        raise AssertionError('a checkpoint with a different synthesis network must be refused')


def check_misc(device):
    """EqualLinear (lr_mul 0.01 + fused lrelu, plain), PixelNorm and minibatch-stddev of the PRODUCT against the reference."""
    from gan_control_amd.models.gan_model import EqualLinear, PixelNorm, minibatch_stddev
    for name, act in (('lin_map', 'fused_lrelu'), ('lin_plain', None)):
        r = group(MISC, name)
        lr_mul = float(r['lr_mul'])
        m = EqualLinear(r['w'].shape[1], r['w'].shape[0], lr_mul=lr_mul, activation=act).to(device)
        with torch.no_grad():
            m.weight.copy_(r['w'])
            m.bias.copy_(r['b'])
            assert rel_err(m(r['x'].to(device)), r['out']) <= TIGHT, name
    r = group(MISC, 'pixel_norm')
    assert rel_err(PixelNorm()(r['x'].to(device)), r['out']) <= TIGHT
    for name in ('mbstd8', 'mbstd2', 'mbstd4'):
        r = group(MISC, name)
        assert rel_err(minibatch_stddev(r['x'].to(device)), r['out']) <= TIGHT, name


def load_configs():
    import json
    import os
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, 'configs.json')) as f:
        return json.load(f)


def check_config_ingestion(device, name='ffhq', size=32, batch=8):
    """The product trainer built from the hot-path fields of a shipped configuration (configs/ffhq.json:5-84 etc.; resolution and
    batch reduced so it runs in seconds): fc_config_from_sub_groups == the reference's MiniBatchUtils.get_fc_config
    (mini_batch_multi_split_utils.py:103-115), the split mapping network has the reference's parameter names, Adam and EMA
    follow generator_trainer.py:161-173, 332, and one full iteration runs."""
    import copy
    from gan_control_amd.trainers.generator_trainer import GeneratorTrainer
    from gan_control_amd.utils.fc_config import fc_config_from_sub_groups
    ref = load_configs()[name]
    fc = fc_config_from_sub_groups(ref['training_config']['sub_groups_dict'], ref['model_config']['latent_size'])
    assert fc.in_order_group_names == ref['fc_config']['in_order_group_names']
    assert {n: {'latent_place': list(g['latent_place']), 'latent_size': g['latent_size']} for n, g in fc.groups.items()} == ref['fc_config']['groups']
    cfg = copy.deepcopy({'model_config': ref['model_config'], 'training_config': ref['training_config']})
    cfg['model_config']['size'] = size
    cfg['training_config']['batch'] = cfg['training_config']['mini_batch'] = batch
    tr = GeneratorTrainer(cfg, device=device, seed=0, fused_adam=False)
    names = [n for n, _ in tr.generator.named_parameters() if n.startswith('style.')]
    assert sorted({n.split('.')[1] for n in names}) == sorted(fc.in_order_group_names)
    for n in fc.in_order_group_names:
        w = dict(tr.generator.named_parameters())
        width = fc.groups[n]['latent_size']
        assert w[f'style.{n}.1.weight'].shape == (256, width) and w[f'style.{n}.8.weight'].shape == (width, 256)
    tc = ref['training_config']
    gr, dr = tc['g_reg_every'] / (tc['g_reg_every'] + 1), tc['d_reg_every'] / (tc['d_reg_every'] + 1)
    pg, pd = tr.g_optim.param_groups[0], tr.d_optim.param_groups[0]
    assert abs(pg['lr'] - tc['lr_g'] * gr) < 1e-12 and pg['betas'] == (0 ** gr, 0.99 ** gr)
    assert abs(pd['lr'] - tc['lr_d'] * dr) < 1e-12 and pd['betas'] == (0 ** dr, 0.99 ** dr)
    assert abs(tr.accum - 0.5 ** (batch / tc['g_moving_average'])) < 1e-15
    assert tr.ada.cfg['enabled'] == tc['augment']['enabled']
    tr.train_iteration(0, tr.synthetic_batch())
    stats = tr.reduced_stats()
    for k in ('d_loss', 'd_r1_loss', 'g_adv_loss', 'g_path_loss'):
        assert stats[k] == stats[k] and abs(stats[k]) < 1e6, (k, stats[k])
    return tr


def check_resblock_blur_adjoint_fusion(device, size=128, batch=2, tol=1e-5):
    """ResBlock with conv1's activation backward folded into the Blur adjoint (gc_upfirdn2d_mask_f32; models/gan_model.py::_FUSE_BLUR_ADJ)
    against the two-launch form: logits, every parameter gradient of a plain backward, the input gradient, and R1's double backward."""
    from gan_control_amd.models import gan_model as gm
    from oracle.networks import procedural_fill_
    torch.manual_seed(0)
    d = gm.Discriminator(size, channel_multiplier=2)
    d.load_state_dict(procedural_fill_(d.state_dict()))
    d = d.to(device)
    gen = torch.Generator().manual_seed(12)
    img = (torch.rand(batch, 3, size, size, generator=gen) * 2 - 1).to(device)
    res = {}
    keep = gm._FUSE_BLUR_ADJ
    try:
        for fused in (True, False):
            gm._FUSE_BLUR_ADJ = fused
            d.zero_grad()
            x = img.clone().requires_grad_(True)
            pred, _ = d(x)
            torch.nn.functional.softplus(pred).mean().backward()
            plain = ({n: p.grad.clone() for n, p in d.named_parameters()}, x.grad.clone(), pred.detach().clone())
            d.zero_grad()
            x = img.clone().requires_grad_(True)
            pred, _ = d(x)
            g, = autograd.grad(pred.sum(), x, create_graph=True)
            g.pow(2).reshape(batch, -1).sum(1).mean().backward()
            r1 = {n: (None if p.grad is None else p.grad.clone()) for n, p in d.named_parameters()}
            res[fused] = (plain, r1, g.detach().clone())
    finally:
        gm._FUSE_BLUR_ADJ = keep
    (pa, xa, la), ra, ga = res[True]
    (pb, xb, lb), rb, gb = res[False]
    assert torch.equal(la, lb)
    assert rel_err(xa, xb) <= tol and rel_err(ga, gb) <= tol
    for n in pb:
        assert rel_err(pa[n], pb[n]) <= tol, n
        assert (ra[n] is None) == (rb[n] is None), n
        if rb[n] is not None and float(rb[n].abs().max()) > 0:
            assert rel_err(ra[n], rb[n]) <= 10 * tol, n


def check_scale_grads_from_sample_wgrad(device, size=64, batch=3, tol=1e-5, pl_tol=None, dtype=torch.float32, thin_only=False, level=1):
    """The modulation / demodulation gradients taken from the per-sample weight gradient (models/op/modulated_conv.py::_samples_route,
    gc_conv2d_wgrad_samples_* + gc_wgrad_samples_contract_f32) against the plane-product route: every parameter gradient of a generator
    backward pass (plain, up-sampling and ToRGB layers all take the route when forced) and of the path-length step's second backward.
    level 1 (the default of the package): nodes of the forward pass only; level 2: also the input-gradient nodes a second backward meets.
    In fp64 (CPU emulation) the routes agree to rounding at both levels.  In split-bf16 arithmetic level 2 does NOT hold up in the
    path-length step: the scale gradient of an input-gradient node cancels exactly against its _PlaneDot partner's (d/dsi of
    sum_p x * (si * g) / si is zero), which the plane route preserves by reading the same tensor twice and a different arithmetic for one
    of the two terms does not (tools/samples_route_probe.py: 5.7e-2 against 2.8e-3 on one layer) -- hence level 1."""
    from gan_control_amd.models import gan_model as gm
    from gan_control_amd.models.op import modulated_conv as mc
    from oracle.networks import procedural_fill_
    torch.manual_seed(0)
    g = gm.Generator(size, 64, 2, channel_multiplier=2, conv_transpose=True)
    g.load_state_dict(procedural_fill_(g.state_dict()))
    g = g.to(device=device, dtype=dtype)
    gen = torch.Generator().manual_seed(3)
    z = torch.randn(batch, 64, generator=gen).to(device=device, dtype=dtype)
    probe = torch.randn(batch, 3, size, size, generator=gen).to(device=device, dtype=dtype)
    keep = (mc._WGRAD_SAMPLES, mc._SAMPLES_MIN_RATIO)
    res, taken = {}, {}
    be = mc._backend.get()
    orig = be.conv2d_wgrad_samples
    try:
        for fused in (True, False):
            mc._WGRAD_SAMPLES, mc._SAMPLES_MIN_RATIO = (level if fused else 0), 0.0
            calls = []
            be.conv2d_wgrad_samples = lambda *a, _c=calls, **k: (_c.append(a[4]), orig(*a, **k))[1]
            g.zero_grad()
            img, _ = g([z], randomize_noise=False)
            (img * probe).sum().backward()
            plain = {n: p.grad.clone() for n, p in g.named_parameters() if p.grad is not None}
            g.zero_grad()
            img, latents = g([z], return_latents=True, randomize_noise=False)
            grad, = autograd.grad((img * probe).sum() / size, latents, create_graph=True)
            grad.pow(2).sum(2).mean(1).sqrt().mean().backward()
            pl = {n: p.grad.clone() for n, p in g.named_parameters() if p.grad is not None}
            res[fused], taken[fused] = (plain, pl), calls
    finally:
        mc._WGRAD_SAMPLES, mc._SAMPLES_MIN_RATIO = keep
        del be.conv2d_wgrad_samples
    assert not taken[False]
    kinds = {(geom.kh, geom.up, geom.down) for geom in taken[True]}
    if thin_only:
        assert kinds == {(1, 1, 1)}, kinds             # fp32 arithmetic on the GPU: only the ToRGB class has the per-sample form
    else:
        assert {(3, 1, 1), (3, 1, 2)} <= kinds, kinds       # plain 3x3 and the swapped form of the up-sampling layers; 1x1 where the backend has the thin form
    (pa, la), (pb, lb) = res[True], res[False]
    assert pa.keys() == pb.keys() and la.keys() == lb.keys()
    for n in pb:
        assert rel_err(pa[n], pb[n]) <= tol, ('backward', n, rel_err(pa[n], pb[n]))
    for n in lb:
        if float(lb[n].abs().max()) > 0:
            assert rel_err(la[n], lb[n]) <= (pl_tol or 10 * tol), ('path length', n, rel_err(la[n], lb[n]))
    return kinds


def check_torgb_fork(device, size=64, batch=2, tol=1e-5, pl_tol=None, dtype=torch.float32):
    """ToRGB handing its input on to the next up-sampling layer (gan_model._FORK_TORGB: the two gradients of a StyledConv output meet in the
    epilogue of ToRGB's input-gradient kernel) against autograd's own sum: image, every parameter gradient, path-length step."""
    from gan_control_amd.models import gan_model as gm
    from oracle.networks import procedural_fill_
    torch.manual_seed(0)
    g = gm.Generator(size, 64, 2, channel_multiplier=2, conv_transpose=True)
    g.load_state_dict(procedural_fill_(g.state_dict()))
    g = g.to(device=device, dtype=dtype)
    gen = torch.Generator().manual_seed(4)
    z = torch.randn(batch, 64, generator=gen).to(device=device, dtype=dtype)
    probe = torch.randn(batch, 3, size, size, generator=gen).to(device=device, dtype=dtype)
    keep, res = gm._FORK_TORGB, {}
    try:
        for fork in (True, False):
            gm._FORK_TORGB = fork
            g.zero_grad()
            img, _ = g([z], randomize_noise=False)
            (img * probe).sum().backward()
            plain = {n: p.grad.clone() for n, p in g.named_parameters() if p.grad is not None}
            g.zero_grad()
            img2, latents = g([z], return_latents=True, randomize_noise=False)
            grad, = autograd.grad((img2 * probe).sum() / size, latents, create_graph=True)
            grad.pow(2).sum(2).mean(1).sqrt().mean().backward()
            res[fork] = (img.detach().clone(), plain, {n: p.grad.clone() for n, p in g.named_parameters() if p.grad is not None})
    finally:
        gm._FORK_TORGB = keep
    (ia, pa, la), (ib, pb, lb) = res[True], res[False]
    assert torch.equal(ia, ib)
    assert pa.keys() == pb.keys() and la.keys() == lb.keys()
    for n in pb:
        assert rel_err(pa[n], pb[n]) <= tol, ('backward', n, rel_err(pa[n], pb[n]))
    for n in lb:
        if float(lb[n].abs().max()) > 0:
            assert rel_err(la[n], lb[n]) <= (pl_tol or 10 * tol), ('path length', n, rel_err(la[n], lb[n]))
