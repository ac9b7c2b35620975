"""INTEGRATION.md sections 2-3 executed: the three-name binding on the reference's OWN gan_model.py and non_leaking.py.

Runs only where /root/reference exists (the build container; skipped on the GPU box, which never sees the reference).  The
reference's modules are imported unmodified; the binding a maintainer adds -- the `if FUSED:` branch importing FusedLeakyReLU,
fused_leaky_relu and upfirdn2d from gan_control_amd.models.op (gan_model.py:19-50), and the package `gan_control.models.op` that
non_leaking.py:6 imports from -- is applied by assigning those module globals, the C ABI underneath is the emulated one (CPU), and
the patched run is compared with the unpatched FUSED=False run: images, logits, every parameter gradient, the None-gradient sets,
and `augment()` with its input gradient.
"""
import os
import sys
import types

import pytest
import torch
from torch import autograd

from conftest import rel_err

REF_SRC = '/root/reference/src'
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_SRC), reason='the reference checkout exists in the build container only')


def _ref_gan_model():
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    import gan_control.models.gan_model as ref_gm
    return ref_gm


def _build(ref_gm, size, state=None):
    from oracle.networks import procedural_fill_
    torch.manual_seed(0)
    g = ref_gm.Generator(size, 512, 8, channel_multiplier=2, conv_transpose=True)
    d = ref_gm.Discriminator(size, channel_multiplier=2)
    if state is None:
        state = (procedural_fill_(g.state_dict()), procedural_fill_(d.state_dict()))
    g.load_state_dict(state[0])
    d.load_state_dict(state[1])
    return g, d, state


def _run(g, d, z, noise):
    for m in (g, d):
        m.zero_grad()
        for p in m.parameters():
            p.requires_grad_(True)
    img, _ = g([z], noise=noise)
    logits, _ = d(img)
    torch.nn.functional.softplus(-logits).mean().backward()
    grads = {('g', n): p.grad for n, p in g.named_parameters()}
    grads.update({('d', n): p.grad for n, p in d.named_parameters()})
    return img.detach(), logits.detach(), grads


def test_three_name_binding_on_the_reference_modules(emu_backend):
    import op_checks as oc
    from gan_control_amd.models import op as amd_op
    ref_gm = _ref_gan_model()
    size, batch = 32, 2
    gen = torch.Generator().manual_seed(9)
    z = torch.randn(batch, 512, generator=gen)
    noise = oc.seeded_noise(size, batch, 21)
    assert ref_gm.FUSED is False
    g, d, state = _build(ref_gm, size)
    img0, logits0, grads0 = _run(g, d, z, noise)
    # --- the binding of INTEGRATION.md section 2: three module globals, models rebuilt so that FusedLeakyReLU instances are ours
    keep = {n: getattr(ref_gm, n) for n in ('FusedLeakyReLU', 'fused_leaky_relu', 'upfirdn2d')}
    try:
        ref_gm.FusedLeakyReLU, ref_gm.fused_leaky_relu, ref_gm.upfirdn2d = amd_op.FusedLeakyReLU, amd_op.fused_leaky_relu, amd_op.upfirdn2d
        g1, d1, _ = _build(ref_gm, size, state)
        assert type(g1.conv1.activate) is amd_op.FusedLeakyReLU
        assert sorted(g1.state_dict()) == sorted(g.state_dict()) and sorted(d1.state_dict()) == sorted(d.state_dict())
        img1, logits1, grads1 = _run(g1, d1, z, noise)
    finally:
        for n, v in keep.items():
            setattr(ref_gm, n, v)
    assert rel_err(img1, img0) < 1e-5 and rel_err(logits1, logits0) < 1e-5
    assert {k for k, v in grads0.items() if v is None} == {k for k, v in grads1.items() if v is None}
    for k, v in grads0.items():
        if v is not None:
            assert rel_err(grads1[k], v) < 2e-5, k
    # second order through the bound operators: R1 on the reference's own Discriminator
    real = (torch.rand(batch, 3, size, size, generator=gen) * 2 - 1)
    out = []
    for disc in (d, d1):
        disc.zero_grad()
        x = real.clone().requires_grad_(True)
        pred, _ = disc(x)
        gr, = autograd.grad(pred.sum(), x, create_graph=True)
        gr.pow(2).reshape(batch, -1).sum(1).mean().backward()
        out.append({n: p.grad for n, p in disc.named_parameters()})
    for n, v in out[0].items():
        assert (v is None) == (out[1][n] is None), n
        if v is not None and float(v.abs().max()) > 0:
            assert rel_err(out[1][n], v) < 5e-5, n


def test_non_leaking_augment_through_the_op_package_stub(emu_backend):
    """non_leaking.py:6 does `from gan_control.models.op import upfirdn2d`: provide that package with this repo's operator (INTEGRATION.md
    section 2, second snippet) and compare the reference's own augment() with the run on the reference's native upfirdn2d."""
    from gan_control_amd.models import op as amd_op
    ref_gm = _ref_gan_model()
    current = {'fn': ref_gm.upfirdn2d}
    stub = types.ModuleType('gan_control.models.op')
    stub.upfirdn2d = lambda *a, **k: current['fn'](*a, **k)
    stub.FusedLeakyReLU, stub.fused_leaky_relu = amd_op.FusedLeakyReLU, amd_op.fused_leaky_relu
    had = sys.modules.get('gan_control.models.op')
    sys.modules['gan_control.models.op'] = stub
    sys.modules.pop('gan_control.trainers.non_leaking', None)
    try:
        import importlib
        nl = importlib.import_module('gan_control.trainers.non_leaking')
        gen = torch.Generator().manual_seed(4)
        img = torch.rand(2, 3, 48, 48, generator=gen) * 2 - 1
        go = torch.randn(2, 3, 48, 48, generator=gen)
        for seed in range(30):            # a transform the reference can pad for (it loops for ever on the others, non_leaking.py:288-313)
            torch.manual_seed(seed)
            G = nl.sample_affine(0.8, 2, 48, 48)
            C = nl.sample_color(0.8, 2)
            pads = nl.get_padding(torch.inverse(G), 48, 48)
            if max(pads) + 6 < 48:
                break
        res = []
        for fn in (ref_gm.upfirdn2d, amd_op.upfirdn2d):
            current['fn'] = fn
            x = img.clone().requires_grad_(True)
            out, _ = nl.augment(x, 0.8, (G, C))
            gi, = autograd.grad(out, x, go[:, :, :out.shape[2], :out.shape[3]])
            res.append((out.detach(), gi))
        assert rel_err(res[1][0], res[0][0]) < 2e-5 and rel_err(res[1][1], res[0][1]) < 2e-5
    finally:
        sys.modules.pop('gan_control.trainers.non_leaking', None)
        if had is None:
            sys.modules.pop('gan_control.models.op', None)
        else:
            sys.modules['gan_control.models.op'] = had
