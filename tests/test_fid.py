"""FID plumbing: the Frechet distance against the reference's own calc_fid outputs (tests/golden/fid.npz),
the oracle restatement against the same vectors, and the generator sampling loop."""
import os
import pickle

import numpy as np
import pytest
import torch

from conftest import load_golden

from gan_control_amd.fid_utils import fid as pfid
from oracle import fid as ofid

GOLD = load_golden('fid')
CASES = sorted({k.split('/')[0] for k in GOLD if k.startswith('case')})


@pytest.mark.parametrize('case', CASES)
def test_frechet_distance_matches_reference_vectors(case):
    m1, c1, m2, c2, want = (GOLD[f'{case}/{k}'] for k in ('m1', 'c1', 'm2', 'c2', 'fid'))
    got = pfid.calc_fid(m1, c1, m2, c2)
    assert abs(got - want) <= 1e-9 * max(1.0, abs(want))
    ora = ofid.frechet_distance(m1, c1, m2, c2)
    assert abs(ora - want) <= 1e-6 * max(1.0, abs(want))


def test_frechet_distance_properties():
    m, c = GOLD['case1/m1'], GOLD['case1/c1']
    assert abs(pfid.calc_fid(m, c, m, c)) < 1e-6 * np.trace(c)
    assert abs(ofid.frechet_distance(m, c, m, c)) < 1e-6 * np.trace(c)
    # symmetric in its two distributions, and a pure mean shift adds |shift|^2
    a = pfid.calc_fid(GOLD['case2/m1'], GOLD['case2/c1'], GOLD['case2/m2'], GOLD['case2/c2'])
    b = pfid.calc_fid(GOLD['case2/m2'], GOLD['case2/c2'], GOLD['case2/m1'], GOLD['case2/c1'])
    assert abs(a - b) <= 1e-7 * abs(a)
    shift = np.full_like(m, 0.25)
    assert abs(pfid.calc_fid(m + shift, c, m, c) - shift @ shift) < 1e-6 * np.trace(c)
    # a singular product takes the eps-ridge branch and stays finite
    z = np.zeros_like(c)
    assert np.isfinite(pfid.calc_fid(m, z, m, z))


@pytest.mark.parametrize('key', [k for k in GOLD if k.startswith('plan/')])
def test_batch_plan(key):
    n, b = (int(v) for v in key.split('/')[1].split('_'))
    assert pfid._batch_plan(n, b) == GOLD[key].tolist() == ofid.batch_plan(n, b)


class _FakeG(torch.nn.Module):
    """Generator stand-in with the reference's call contract: g([z]) -> (img, None)."""

    def __init__(self, channels):
        super().__init__()
        self.channels, self.calls = channels, []

    def forward(self, styles):
        z = styles[0]
        self.calls.append(z.shape[0])
        return z[:, :self.channels * 16].reshape(z.shape[0], self.channels, 4, 4), None


def _feature_net(img):
    return (img.mean((2, 3), keepdim=True),)


def test_sample_features_and_evaluate(tmp_path):
    g = _FakeG(3)
    torch.manual_seed(0)
    f = pfid.sample_features(g, _feature_net, 8, 21, device='cpu')
    assert f.shape == (21, 3) and g.calls == [8, 8, 5]
    g1 = _FakeG(1)
    f1 = pfid.sample_features(g1, _feature_net, 4, 8, device='cpu')
    assert f1.shape == (8, 3) and torch.equal(f1[:, 0], f1[:, 2])        # single-channel images are replicated
    mean, cov = pfid.feature_statistics(f.numpy())
    stat = tmp_path / 'stats.pkl'
    with open(stat, 'wb') as fh:
        pickle.dump({'mean': mean, 'cov': cov}, fh)
    torch.manual_seed(0)
    fid = pfid.evaluate_fid(_FakeG(3), _feature_net, 8, 21, 'cpu', str(stat))
    assert abs(fid) < 1e-8                                                 # same seed -> same samples -> distance 0


@pytest.mark.gpu
def test_sample_features_through_the_hip_generator():
    from gan_control_amd.models.gan_model import Generator
    torch.manual_seed(3)
    g = Generator(32, 512, 2, channel_multiplier=2, conv_transpose=True, fc_config=None).cuda().eval()
    pool = torch.nn.AdaptiveAvgPool2d(4)
    torch.manual_seed(1)
    a = pfid.sample_features(g, lambda img: (pool(img),), 4, 10)
    torch.manual_seed(1)
    b = pfid.sample_features(g, lambda img: (pool(img),), 4, 10)
    assert a.shape == (10, 48) and torch.isfinite(a).all()
    ma, ca = pfid.feature_statistics(a.numpy())
    mb, cb = pfid.feature_statistics(b.numpy())
    # noise injection is random per call: statistics are close, not identical; the distance is small next to the feature scale
    assert pfid.calc_fid(ma, ca, mb, cb) < 0.5 * (np.trace(ca) + 1e-6)


@pytest.mark.gpu
def test_fid_of_the_composed_hip_pipeline_against_the_oracle():
    """BASELINE's second metric end to end, on procedural weights (the FID checkpoint and the real-image statistics are external
    downloads): HIP generator -> HIP InceptionV3 (299 x 299 resize, pool3) -> feature statistics -> calc_fid between two seeded sample
    sets of 200 images each, against the same number reached through oracle/networks.py -> oracle/inception.py -> oracle/fid.py on the
    same latents and noise maps on the host.  The test prints the measured errors; asserted: 1e-3 on the features (the bound of
    tests/test_inception.py for the HIP feature network) and 1e-3 on the distance."""
    import math
    from gan_control_amd.models.gan_model import Generator
    from gan_control_amd.fid_utils.inception import InceptionV3
    from oracle import networks, inception as oinc
    size, n_per_set, batch = 64, 200, 40
    g = Generator(size, 512, 8, channel_multiplier=2, conv_transpose=True)
    g_sd = networks.procedural_fill_(g.state_dict())
    g.load_state_dict(g_sd)
    g = g.cuda().eval()
    net = InceptionV3(output_blocks=[3], normalize_input=False)              # the reference's evaluation call (tracker.py:322-329)
    i_sd = oinc.procedural_inception_fill_(net.state_dict())
    net.load_state_dict(i_sd)
    net = net.cuda().eval()
    gen = torch.Generator().manual_seed(50)
    num_layers = (int(math.log2(size)) - 2) * 2 + 1
    plan = pfid._batch_plan(n_per_set, batch)
    sets = []
    for s in range(2):
        zs = [torch.randn(b, 512, generator=gen) * (1.0 if s == 0 else 0.7) for b in plan]       # the second set is a narrower distribution
        noises = [[torch.randn(b, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2), generator=gen) for i in range(num_layers)] for b in plan]
        sets.append((zs, noises))
    fids = {}
    feats = {}
    for side in ('hip', 'oracle'):
        stats = []
        for s, (zs, noises) in enumerate(sets):
            it = iter(range(len(plan)))

            def gen_fn(z_unused, zs=zs, noises=noises, it=it):
                k = next(it)
                if side == 'hip':
                    return g([zs[k].cuda()], noise=[n.cuda() for n in noises[k]])[0]
                with torch.no_grad():
                    return networks.generator_forward(g_sd, [zs[k]], size, noise=noises[k])[0]

            if side == 'hip':
                f = pfid.sample_features(g, net, batch, n_per_set, device='cuda', generator_fn=gen_fn)
            else:
                fnet = lambda img: oinc.inception_features(i_sd, img, (3,), True, False)
                f = pfid.sample_features(None, fnet, batch, n_per_set, device='cpu', generator_fn=gen_fn)
            assert f.shape == (n_per_set, 2048)
            feats[side, s] = f
            stats.append(pfid.feature_statistics(f.numpy()))
        (m1, c1), (m2, c2) = stats
        fids[side] = pfid.calc_fid(m1, c1, m2, c2) if side == 'hip' else ofid.frechet_distance(m1, c1, m2, c2)
    e_feat = max(float((feats['hip', s] - feats['oracle', s]).abs().max() / feats['oracle', s].abs().max()) for s in range(2))
    e_fid = abs(fids['hip'] - fids['oracle']) / abs(fids['oracle'])
    print('composed FID: hip %.6f oracle %.6f (rel %.2e); features rel %.2e' % (fids['hip'], fids['oracle'], e_fid, e_feat))
    assert fids['oracle'] > 0 and e_feat < 1e-3 and e_fid < 1e-3, (fids, e_feat, e_fid)
