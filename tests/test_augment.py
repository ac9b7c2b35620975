"""ADA augmentation: oracle vs golden, product matrix sampling vs golden (same RNG call order),
product image path vs golden on the emulated C ABI (CPU) and on the HIP kernels (GPU)."""
import numpy as np
import pytest
import torch
from torch import autograd

from conftest import load_golden, group, rel_err

AUG = load_golden('augment')


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_oracle_augment(tag):
    from oracle import augment as oaug
    from gan_control_amd.trainers.non_leaking import SYM6
    r = group(AUG, tag)
    assert rel_err(oaug.augment(r['img'], r['G'], r['C'], SYM6), r['out']) < 1e-5


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_sampled_matrices_match_reference_rng_order(tag):
    from gan_control_amd.trainers import non_leaking as nl
    r = group(AUG, tag)
    b, h, w, seed = [int(v) for v in r['cfg']]
    torch.manual_seed(seed)
    G = nl.sample_affine(float(r['p']), b, h, w)
    C = nl.sample_color(float(r['p']), b)
    assert torch.allclose(G, r['G'], atol=1e-6) and torch.allclose(C, r['C'], atol=1e-6)


def _check_product(tag, device):
    from gan_control_amd.trainers import non_leaking as nl
    r = group(AUG, tag)
    x = r['img'].to(device).requires_grad_(True)
    out, (G, C) = nl.augment(x, float(r['p']), (r['G'], r['C']))
    assert out.shape == r['out'].shape
    assert rel_err(out, r['out']) < 2e-5
    gi, = autograd.grad(out, x, r['go'].to(device))
    assert rel_err(gi, r['gi']) < 2e-5


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_product_augment_emulated(tag, emu_backend):
    _check_product(tag, 'cpu')


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['a', 'b'])
def test_product_augment_gpu(tag):
    _check_product(tag, 'cuda')


def test_ada_controller():
    from gan_control_amd.trainers.non_leaking import AdaptiveAugmentState
    st = AdaptiveAugmentState({'enabled': True, 'ada_target': 0.6, 'ada_length': 1000, 'p': 0}, 'cpu')
    for _ in range(63):
        st.update(torch.ones(4, 1))
    assert st.p == 0.0 and float(st.accum[1]) == 252
    p = st.update(torch.ones(4, 1))                      # 256 predictions, all positive: r_t = 1 > target
    assert abs(p - 0.6 / 1000 * 256) < 1e-9 and float(st.accum[1]) == 0 and st.r_t == 1.0
    for _ in range(64):
        p = st.update(-torch.ones(4, 1))                 # r_t = -1 < target: p goes back down, clamped at 0
    assert p == 0.0
