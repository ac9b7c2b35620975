"""FID: generator sampling, the InceptionV3 feature network on the HIP kernels and the Fréchet distance.

The pretrained weights of the feature network and the real-image statistics are external assets (inception.py:14 downloads the
weights, inception_stats/*.pkl is not shipped): ``InceptionV3.load_fid_weights`` takes the downloaded checkpoint's state dict, and any
callable with the reference's contract -- ``net(img)[0]`` is a [B, F, ...] feature tensor -- can be plugged in instead.
"""
from .fid import sample_features, frechet_distance, calc_fid, evaluate_fid, feature_statistics
from .inception import InceptionV3

__all__ = ['sample_features', 'frechet_distance', 'calc_fid', 'evaluate_fid', 'feature_statistics', 'InceptionV3']
