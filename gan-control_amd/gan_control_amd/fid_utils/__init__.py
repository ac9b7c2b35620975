"""FID plumbing: generator sampling side and the Fréchet distance.

The InceptionV3 feature network and the real-image statistics are external assets (inception.py:14 downloads the
weights, inception_stats/*.pkl is not shipped); any callable with the reference's contract -- ``net(img)[0]`` is a
[B, F, ...] feature tensor -- can be plugged in.
"""
from .fid import sample_features, frechet_distance, calc_fid, evaluate_fid, feature_statistics

__all__ = ['sample_features', 'frechet_distance', 'calc_fid', 'evaluate_fid', 'feature_statistics']
