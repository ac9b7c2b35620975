"""Generator-side FID evaluation.

Reference: fid_utils/fid.py:14-40 (extract_feature_from_samples), :43-66 (calc_fid) and evaluate_fid.py:11-35.
The generator forward is the HIP hot path (gan_control_amd.models.gan_model.Generator); the distance itself is
host-side numpy / scipy exactly as in the reference (it runs once per 10 000 iterations on 2048-dimensional statistics).
"""
import pickle
import time

import numpy as np
import torch
from scipy import linalg


def _batch_plan(n_sample, batch_size):
    """Full batches followed by one remainder batch (fid.py:22-27)."""
    full, rest = divmod(int(n_sample), int(batch_size))
    return [int(batch_size)] * full + ([rest] if rest else [])


@torch.no_grad()
def sample_features(generator, feature_net, batch_size, n_sample, device='cuda', training=True, latent_dim=512, generator_fn=None):
    """Features of ``n_sample`` generated images, on the host, in generation order.

    ``feature_net(img)[0]`` is flattened per image (fid.py:34); single-channel images are replicated to three channels
    (fid.py:32-33).  ``training`` only silences the progress output in the reference and is accepted for call compatibility.
    """
    feats = []
    for b in _batch_plan(n_sample, batch_size):
        z = torch.randn(b, latent_dim, device=device)
        img = generator([z])[0] if generator_fn is None else generator_fn(z)
        if img.shape[1] == 1:
            img = img.expand(-1, 3, -1, -1)
        f = feature_net(img)[0]
        feats.append(f.reshape(img.shape[0], -1).to('cpu'))
    return torch.cat(feats, 0)


def feature_statistics(features):
    """Mean and (unbiased, as np.cov) covariance of a [n, F] feature matrix (evaluate_fid.py:25-26)."""
    f = np.asarray(features, dtype=np.float64) if not isinstance(features, np.ndarray) else features
    return np.mean(f, 0), np.cov(f, rowvar=False)


def frechet_distance(sample_mean, sample_cov, real_mean, real_cov, eps=1e-6):
    """|m1 - m2|^2 + tr(C1) + tr(C2) - 2 tr((C1 C2)^(1/2)), with the reference's handling of a singular product
    (retry with eps on both diagonals) and of a complex square root (imaginary diagonal above 1e-3 is an error)."""
    root, _ = linalg.sqrtm(sample_cov @ real_cov, disp=False)
    if not np.isfinite(root).all():
        print('product of cov matrices is singular')
        ridge = eps * np.eye(sample_cov.shape[0])
        root = linalg.sqrtm((sample_cov + ridge) @ (real_cov + ridge))
    if np.iscomplexobj(root):
        if not np.allclose(np.diagonal(root).imag, 0, atol=1e-3):
            raise ValueError(f'Imaginary component {np.max(np.abs(root.imag))}')
        root = root.real
    delta = sample_mean - real_mean
    return delta @ delta + np.trace(sample_cov) + np.trace(real_cov) - 2 * np.trace(root)


calc_fid = frechet_distance        # the reference's name (fid.py:43)


def evaluate_fid(generator, feature_net, batch, n_sample, device, inception_stat_path, training=False):
    """FID of ``n_sample`` generated images against pickled real statistics {'mean', 'cov'} (evaluate_fid.py:11-35)."""
    t0 = time.time()
    generator.eval()
    if hasattr(feature_net, 'eval'):
        feature_net.eval()
    feats = sample_features(generator, feature_net, batch, n_sample, device, training=training).numpy()
    print(f'extracted {feats.shape[0]} features')
    mean, cov = feature_statistics(feats)
    with open(inception_stat_path, 'rb') as f:
        real = pickle.load(f)
    fid = frechet_distance(mean, cov, real['mean'], real['cov'])
    print('fid: %.3f, time: %.3f (min)' % (fid, (time.time() - t0) / 60))
    return fid
