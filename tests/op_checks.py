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
