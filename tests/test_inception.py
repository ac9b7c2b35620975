"""f-2: the FID feature network.  The oracle (oracle/inception.py) is pinned to the reference's own classes by
tests/golden/inception.npz (oracle/make_golden.py::golden_inception); the product (fid_utils/inception.py) is checked against the same
fixture through the emulated C ABI (CPU) and through the HIP kernels (-m gpu)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import inception as oinc  # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'inception.npz'))
DEV = 'cuda'


def rel_err(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _inputs():
    gen = torch.Generator().manual_seed(21)
    out = {}
    for tag in ('a', 'b'):
        x = torch.rand(tuple(int(v) for v in GOLD[f'{tag}/shape']), generator=gen)
        assert torch.equal(x.flatten()[:64], torch.from_numpy(GOLD[f'{tag}/x_check'])), 'the redrawn input is not the one the fixture was made from'
        out[tag] = x
    return out


def _product(device, output_blocks=(0, 1, 2, 3), **kw):
    from gan_control_amd.fid_utils.inception import InceptionV3
    net = InceptionV3(output_blocks=list(output_blocks), **kw)
    net.load_state_dict(oinc.procedural_inception_fill_(net.state_dict()))
    return net.to(device)


def _check_against_fixture(outs, tag, tol):
    assert rel_err(outs[3].reshape(outs[3].shape[0], -1), GOLD[f'{tag}/pool3']) <= tol
    for i in range(3):
        assert rel_err(outs[i].mean((2, 3)), GOLD[f'{tag}/block{i}_mean']) <= tol, i
        assert rel_err(outs[i][:, ::7, ::5, ::3], GOLD[f'{tag}/block{i}_sample']) <= tol, i


def test_oracle_matches_reference_fixture():
    from gan_control_amd.fid_utils.inception import InceptionV3
    sd = oinc.procedural_inception_fill_(InceptionV3(output_blocks=[0, 1, 2, 3]).state_dict())
    assert len([k for k in sd if not k.endswith('num_batches_tracked')]) == int(GOLD['n_keys'][0]), 'state_dict keys differ from the reference wrapper'
    for tag, x in _inputs().items():
        with torch.no_grad():
            _check_against_fixture(oinc.inception_features(sd, x, (0, 1, 2, 3)), tag, 1e-5)


def test_product_emulated_matches_fixture(emu_backend):
    net = _product('cpu')
    for tag, x in _inputs().items():
        with torch.no_grad():
            _check_against_fixture(net(x), tag, 1e-4)


def test_product_options_and_checkpoint_loading(emu_backend):
    """resize_input / normalize_input off, a single output block, and the un-wrapped checkpoint's key names (load_fid_weights)."""
    from gan_control_amd.fid_utils.inception import InceptionV3
    net = _product('cpu', output_blocks=(3,), resize_input=False, normalize_input=False)
    x = torch.from_numpy(GOLD['c/x'])
    x_big = torch.nn.functional.interpolate(x, size=(96, 80), mode='bilinear', align_corners=False)
    with torch.no_grad():
        out = net(x_big)
    assert len(out) == 1 and rel_err(out[0].reshape(2, -1), GOLD['c/pool3']) <= 1e-4
    # rename the wrapper's keys back to the names of the FID checkpoint and load them through load_fid_weights
    raw = {}
    for k, v in net.state_dict().items():
        parts = k.split('.')
        raw[InceptionV3._LAYOUT[int(parts[1])][int(parts[2])] + '.' + '.'.join(parts[3:])] = v.clone()
    raw['fc.weight'] = torch.zeros(1008, 2048)
    fresh = InceptionV3(output_blocks=[3], resize_input=False, normalize_input=False)
    fresh.load_fid_weights(raw)
    with torch.no_grad():
        assert torch.equal(fresh(x_big)[0], out[0])
    with pytest.raises(NotImplementedError):
        fresh.train()(x_big)


def test_fid_pipeline_with_the_feature_network(emu_backend):
    """extract features -> statistics -> Fréchet distance with the built network in the loop (fid.py:14-66)."""
    from gan_control_amd.fid_utils import feature_statistics, calc_fid
    net = _product('cpu', output_blocks=(3,))
    gen = torch.Generator().manual_seed(3)
    with torch.no_grad():
        f1 = net(torch.rand(6, 3, 32, 32, generator=gen))[0].reshape(6, -1).double().numpy()
        f2 = net(torch.rand(6, 3, 32, 32, generator=gen) * 0.5)[0].reshape(6, -1).double().numpy()
    m1, c1 = feature_statistics(f1)
    m2, c2 = feature_statistics(f2)
    assert calc_fid(m1, c1, m1, c1) < 1e-3 * (1 + abs(float(np.trace(c1))))
    assert calc_fid(m1, c1, m2, c2) > 0


@pytest.mark.gpu
def test_product_hip_matches_fixture():
    net = _product(DEV)
    for tag, x in _inputs().items():
        with torch.no_grad():
            _check_against_fixture([o.cpu() for o in net(x.to(DEV))], tag, 1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(2, 5, 70, 19, 23, 3, 3, 1, 1, 1), (1, 48, 64, 17, 17, 5, 5, 1, 2, 2), (2, 9, 130, 12, 31, 1, 7, 1, 0, 3), (2, 9, 130, 31, 12, 7, 1, 1, 3, 0),
                                  (1, 33, 40, 35, 35, 3, 3, 2, 0, 0), (3, 16, 8, 9, 9, 1, 1, 1, 0, 0), (1, 20, 65, 10, 14, 1, 3, 1, 0, 1), (1, 20, 65, 14, 10, 3, 1, 1, 1, 0)])
def test_inception_conv_kernel(case):
    """gc_conv2d_bn_relu_f32 against ATen in fp64: every tap shape of the network, ragged tiles and channel blocks, channel-offset output."""
    from gan_control_amd.models.op import _backend
    b, k, n, h, w, kh, kw, stride, py, px = case
    gen = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x, wt = torch.randn(b, k, h, w, generator=gen), torch.randn(n, k, kh, kw, generator=gen)
    scale, shift = torch.rand(n, generator=gen) + 0.5, torch.randn(n, generator=gen)
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), wt.double(), None, stride, (py, px)) * scale.double().reshape(1, -1, 1, 1) + shift.double().reshape(1, -1, 1, 1))
    hip = _backend.get()
    out = hip.conv2d_bn_relu(x.to(DEV), wt.to(DEV), scale.to(DEV), shift.to(DEV), stride, py, px, True)
    assert rel_err(out, ref) < 5e-6
    wide = torch.full((b, n + 7, ref.shape[2], ref.shape[3]), -3.0, device=DEV)
    hip.conv2d_bn_relu(x.to(DEV), wt.to(DEV), scale.to(DEV), shift.to(DEV), stride, py, px, True, wide, 4)
    assert torch.equal(wide[:, 4:4 + n], out) and bool((wide[:, :4] == -3).all()) and bool((wide[:, 4 + n:] == -3).all())


@pytest.mark.gpu
def test_inception_pool_and_resize_kernels():
    from gan_control_amd.models.op import _backend
    import torch.nn.functional as F
    hip = _backend.get()
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(2, 7, 19, 23, generator=gen)
    xd = x.to(DEV)
    assert torch.equal(hip.pool2d(xd, 3, 2, 0, 'max').cpu(), F.max_pool2d(x, 3, 2))
    assert torch.equal(hip.pool2d(xd, 3, 1, 1, 'max').cpu(), F.max_pool2d(x, 3, 1, 1))
    assert rel_err(hip.pool2d(xd, 3, 1, 1, 'avg'), F.avg_pool2d(x, 3, 1, 1, count_include_pad=False)) < 1e-6
    assert rel_err(hip.global_avgpool(xd), x.mean((2, 3), keepdim=True)) < 1e-6
    for size in ((299, 299), (40, 31), (19, 23)):
        assert rel_err(hip.resize_bilinear(xd, size[0], size[1], 2.0, -1.0), 2 * F.interpolate(x, size=size, mode='bilinear', align_corners=False) - 1) < 1e-5
