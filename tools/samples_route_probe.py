"""Path-length step (the ill-conditioned one) with the scale gradients from the per-sample weight gradients and from the plane products, each
against the fp64 CPU emulation of the same network: is either route closer to the truth?  Usage: python tools/samples_route_probe.py [size batch mode level]"""
import os
import sys

import torch
from torch import autograd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'gan-control_amd'), os.path.join(ROOT, 'tests')]


def grads(g, z, probe, size):
    out = {}
    g.zero_grad()
    img, _ = g([z], randomize_noise=False)
    (img * probe).sum().backward()
    out['backward'] = {n: p.grad.detach().double().cpu().clone() for n, p in g.named_parameters() if p.grad is not None}
    g.zero_grad()
    img, latents = g([z], return_latents=True, randomize_noise=False)
    grad, = autograd.grad((img * probe).sum() / size, latents, create_graph=True)
    grad.pow(2).sum(2).mean(1).sqrt().mean().backward()
    out['path length'] = {n: p.grad.detach().double().cpu().clone() for n, p in g.named_parameters() if p.grad is not None}
    return out


def main():
    from conftest import EmulatedBackend, rel_err
    from gan_control_amd.models import gan_model as gm
    from gan_control_amd.models.op import _backend, modulated_conv as mc
    from oracle.networks import procedural_fill_
    size, batch = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 2)
    mode = sys.argv[3] if len(sys.argv) > 3 else 'bf16x3'
    level = int(sys.argv[4]) if len(sys.argv) > 4 else 2          # GANCONTROL_WGRAD_SAMPLES: 1 = nodes of the forward pass only
    torch.manual_seed(0)
    g = gm.Generator(size, 64, 2, channel_multiplier=2, conv_transpose=True)
    g.load_state_dict(procedural_fill_(g.state_dict()))
    gen = torch.Generator().manual_seed(3)
    z = torch.randn(batch, 64, generator=gen)
    probe = torch.randn(batch, 3, size, size, generator=gen)
    hip = _backend.get()
    hip.conv_mode = mode
    g = g.cuda()
    res = {}
    for route in (True, False):
        mc._WGRAD_SAMPLES, mc._SAMPLES_MIN_RATIO = (level if route else 0), 0.0
        res[route] = grads(g, z.cuda(), probe.cuda(), size)
    prev = _backend._install_for_tests(EmulatedBackend())
    mc._WGRAD_SAMPLES = 0
    truth = grads(g.cpu().double(), z.double(), probe.double(), size)
    _backend._install_for_tests(prev)
    for step in ('backward', 'path length'):
        print(f'--- {step}: relative error against fp64 (samples route, plane route), route vs route')
        worst = [0.0, 0.0, 0.0]
        for n, t in truth[step].items():
            if float(t.abs().max()) == 0:
                continue
            e = [float(rel_err(res[True][step][n], t)), float(rel_err(res[False][step][n], t)), float(rel_err(res[True][step][n], res[False][step][n]))]
            worst = [max(a, b) for a, b in zip(worst, e)]
            if 'modulation' in n or 'style' in n or e[2] > 1e-3:
                print(f'{n:40s} {e[0]:.2e} {e[1]:.2e} {e[2]:.2e}')
        print(f'{"worst":40s} {worst[0]:.2e} {worst[1]:.2e} {worst[2]:.2e}')


if __name__ == '__main__':
    main()
