"""The operator socket of gan-control, filled with gfx950 kernels.

The reference binds ``FusedLeakyReLU``, ``fused_leaky_relu`` and ``upfirdn2d`` at
gan_model.py:19-50 (and imports ``upfirdn2d`` from ``gan_control.models.op`` at
non_leaking.py:6, a package it does not ship).  This package provides those three names with
identical signatures, plus the convolutions the north star adds behind the same boundary.
"""
from .fused_act import FusedLeakyReLU, fused_leaky_relu, fused_noise_bias_act
from .upfirdn2d import upfirdn2d, upfirdn2d_bias_act
from . import conv2d_gradfix
from .modulated_conv import modulated_conv2d, modulated_conv2d_act, demod_coefficients
from .warp import affine_warp_bilinear, reflect_pad

__all__ = ['FusedLeakyReLU', 'fused_leaky_relu', 'fused_noise_bias_act', 'upfirdn2d', 'upfirdn2d_bias_act', 'conv2d_gradfix',
           'modulated_conv2d', 'modulated_conv2d_act', 'demod_coefficients', 'affine_warp_bilinear', 'reflect_pad']
