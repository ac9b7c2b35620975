"""The grouped dense layers of the style path (op/style.py, gc_grouped_linear_*): values, first and second derivatives against plain
fp64 ATen on the emulated C ABI (CPU) and on the HIP kernels (-m gpu), and the Generator's grouped style path against the per-layer path."""
import os

import pytest
import torch

from conftest import rel_err

from gan_control_amd.models.op import style as st

# (n, k) per group, batch: modulation-like (k = 512), demodulation-like (k = channel counts down to 32), ragged row counts, one group
CASES = [
    ([(512, 512), (256, 512), (32, 512), (3, 512)], 4),
    ([(512, 512), (512, 256), (64, 128), (32, 64), (32, 32)], 2),
    ([(17, 8), (5, 4), (130, 36)], 3),
    ([(64, 512)], 1),
    ([(40, 128), (24, 128)], 9),            # more samples than one register tile
    ([(33, 2048)], 16),
]


def _make(case, device, dtype, gap=False):
    groups, batch = case
    gen = torch.Generator().manual_seed(sum(n + k for n, k in groups) + batch)
    specs, at = [], 0
    for i, (n, k) in enumerate(groups):
        if gap and i == 1:
            at += 8                          # an input block no group reads: its gradient must come back as zeros
        specs.append(st.GroupSpec(n, k, 0.1 + 0.05 * i, 0.5 + 0.1 * i, at))
        at += k
    plan = st.Plan(specs, in_cols=at + (4 if gap else 0))
    x = torch.randn(batch * plan.in_cols, generator=gen).to(device=device, dtype=dtype).requires_grad_(True)
    ws = [(torch.randn(n, k, generator=gen) / k ** 0.5).to(device=device, dtype=dtype).requires_grad_(True) for n, k in groups]
    bs = [(torch.randn(n, generator=gen).to(device=device, dtype=dtype).requires_grad_(True) if i % 2 == 0 else None) for i, (n, _) in enumerate(groups)]
    cot = torch.randn(batch * plan.out_cols, generator=gen).to(device=device, dtype=dtype)
    v = torch.randn(batch * plan.in_cols, generator=gen).to(device=device, dtype=dtype)
    return plan, batch, x, ws, bs, cot, v


def _reference(plan, batch, x, ws, bs):
    out = []
    for sp, w, b in zip(plan.specs, ws, bs):
        y = sp.alpha * (x[batch * sp.xcol: batch * (sp.xcol + sp.k)].view(batch, sp.k) @ w.t())
        out.append((y + sp.beta * b if b is not None else y).reshape(-1))
    return torch.cat(out)


def _derivatives(fn, plan, batch, x, ws, bs, cot, v):
    """value, first derivatives, and the derivatives of <dL/dx, v> + sum <dL/dw, w> (second order through every primitive)."""
    y = fn(x, ws, bs)
    leaves = [x] + ws + [b for b in bs if b is not None]
    first = torch.autograd.grad((y * cot).sum() + (y ** 2).sum(), leaves, create_graph=True)
    probe = (first[0] * v).sum() + sum((g * w.detach()).sum() for g, w in zip(first[1:1 + len(ws)], ws))
    second = torch.autograd.grad(probe, leaves, allow_unused=True)
    return [y] + list(first) + [s for s in second if s is not None]


def _check(case, device, tol, gap=False):
    plan, batch, x, ws, bs, cot, v = _make(case, device, torch.float32, gap)
    got = _derivatives(lambda xx, ww, bb: st.grouped_linear(xx, batch, plan, ww, bb), plan, batch, x, ws, bs, cot, v)
    d = lambda t: None if t is None else t.detach().double().cpu().requires_grad_(True)
    xd, wd, bd = d(x), [d(w) for w in ws], [d(b) for b in bs]
    ref = _derivatives(lambda xx, ww, bb: _reference(plan, batch, xx, ww, bb), plan, batch, xd, wd, bd, cot.double().cpu(), v.double().cpu())
    assert len(got) == len(ref)
    for i, (a, b) in enumerate(zip(got, ref)):
        assert a.shape == b.shape and rel_err(a, b) < tol, (i, rel_err(a, b))


@pytest.mark.parametrize('case', CASES[:4])
def test_grouped_linear_autograd_emulated(case, emu_backend):
    _check(case, 'cpu', 2e-5)


def test_grouped_linear_unread_input_blocks_emulated(emu_backend):
    _check(CASES[1], 'cpu', 2e-5, gap=True)


@pytest.mark.gpu
@pytest.mark.parametrize('case', CASES)
def test_grouped_linear_hip(case):
    _check(case, 'cuda', 2e-5)
    _check(case, 'cuda', 2e-5, gap=True)


@pytest.mark.gpu
def test_grouped_linear_hip_is_deterministic_and_matches_per_layer_calls():
    plan, batch, x, ws, bs, cot, v = _make(CASES[0], 'cuda', torch.float32)
    a = st.grouped_linear(x, batch, plan, ws, bs)
    assert torch.equal(a, st.grouped_linear(x, batch, plan, ws, bs))
    from gan_control_amd.models.op.linear import equal_linear, scaled_mm
    at = 0
    for sp, w, b in zip(plan.specs, ws, bs):
        xb = x[batch * sp.xcol: batch * (sp.xcol + sp.k)].view(batch, sp.k)
        ref = equal_linear(xb, w, b, sp.alpha, sp.beta) if b is not None else scaled_mm(xb, w.t(), sp.alpha)
        assert rel_err(a[batch * at: batch * (at + sp.n)].view(batch, sp.n), ref) < 2e-6
        at += sp.n


def test_more_groups_than_one_launch_table_emulated(emu_backend):
    _check(([(8, 16)] * 40, 2), 'cpu', 2e-5)


@pytest.mark.gpu
def test_more_groups_than_one_launch_table_hip():
    _check(([(8, 16)] * 40, 2), 'cuda', 2e-5)        # 40 groups: two launches of the 32-entry table


def _generator_pair(device, size=32):
    """The same generator evaluated through the grouped style path and through the per-layer path."""
    from gan_control_amd.models import gan_model
    from oracle.networks import procedural_fill_
    torch.manual_seed(0)
    g = gan_model.Generator(size, 512, 8, channel_multiplier=2, conv_transpose=True)
    g.load_state_dict(procedural_fill_(g.state_dict()))
    return g.to(device), gan_model


def _style_paths_agree(device, tol):
    import op_checks as oc
    g, gm = _generator_pair(device)
    gen = torch.Generator().manual_seed(6)
    z = [torch.randn(3, 512, generator=gen).to(device), torch.randn(3, 512, generator=gen).to(device)]
    noise = oc.seeded_noise(32, 3, 4, device)
    out = {}
    for fused in (True, False):
        gm._FUSED_STYLE = fused
        try:
            g.zero_grad()
            img, lat = g(z, noise=noise, return_latents=True, inject_index=3)
            gl = gm.Generator.g_path_regularize_grad(img, lat, pl_noise=torch.ones_like(img))
            pen = gl.pow(2).sum(2).mean(1).sqrt().sub(0.3).pow(2).mean()
            (pen + img.square().mean()).backward()
            out[fused] = (img.detach(), gl.detach(), {n: p.grad.clone() for n, p in g.named_parameters() if p.grad is not None})
        finally:
            gm._FUSED_STYLE = True
    assert rel_err(out[True][0], out[False][0]) < tol and rel_err(out[True][1], out[False][1]) < tol
    assert sorted(out[True][2]) == sorted(out[False][2])
    for n, v in out[False][2].items():
        assert rel_err(out[True][2][n], v) < 10 * tol, n


def test_generator_grouped_style_path_equals_per_layer_path_emulated(emu_backend):
    _style_paths_agree('cpu', 2e-5)


@pytest.mark.gpu
def test_generator_grouped_style_path_equals_per_layer_path_hip():
    _style_paths_agree('cuda', 2e-5)
