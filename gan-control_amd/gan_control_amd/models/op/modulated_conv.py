"""Weight-(de)modulated convolution without per-sample weights.

Reference: ModulatedConv2d.forward gan_model.py:281-331 materialises ``[B*OC, IC, k, k]`` weights
and runs a grouped conv with groups = B.  Algebraically the same result is

    y = d[b,oc] * conv(x * s[b,ic], W * scale),   d = rsqrt(scale^2 * sum_ic s^2 * sum_k W^2 + 1e-8)

i.e. ONE shared-weight convolution with a per-(b,ic) scale on the way in and a per-(b,oc) scale
on the way out (relative error vs the reference ~3e-7).
"""
import math

import torch

from . import conv2d_gradfix, _backend
from ._backend import ConvGeom
from .upfirdn2d import upfirdn2d


def demod_coefficients(weight, s, scale, eps=1e-8):
    """d[b,oc] = rsqrt(sum_{ic,k} (scale * W[oc,ic,k] * s[b,ic])^2 + eps); weight is [1,OC,IC,k,k], s is [B,IC]."""
    wsq = weight[0].pow(2).sum([2, 3])                     # [OC, IC]
    return torch.rsqrt((s.pow(2) @ wsq.t()) * (scale * scale) + eps)


def modulated_conv2d(x, weight, s, demodulate=True, upsample=False, blur_kernel=None, blur_pad=None, padding=None):
    """x [B,IC,H,W]; weight [1,OC,IC,k,k] (the reference parameter layout); s [B,IC] = modulation(style).

    plain:    conv2d(padding = k // 2)                                    gan_model.py:325-329
    upsample: conv_transpose2d(stride 2, padding 0) -> Blur(blur_pad)     gan_model.py:295-307
    """
    _, oc, ic, k, _ = weight.shape
    scale = 1.0 / math.sqrt(ic * k * k)
    d = demod_coefficients(weight, s, scale) if demodulate else None
    w = weight[0] * scale                                                  # [OC, IC, k, k]
    be = _backend.get()
    fused = not (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or s.requires_grad))
    if upsample:
        w_t = w.flip(2, 3).permute(2, 3, 1, 0).contiguous()                # correlation form, [k,k,IC,OC]
        if fused:
            oh, ow = (x.shape[2] - 1) * 2 + k, (x.shape[3] - 1) * 2 + k
            y = be.conv2d(x.contiguous(), w_t, s.contiguous(), None if d is None else d.contiguous(),
                          ConvGeom(k, k, 2, 1, k - 1, k - 1, oh, ow))
        else:
            y = conv2d_gradfix.conv_transpose2d_t(x * s[:, :, None, None], w_t, stride=2, padding=0)
            if d is not None:
                y = y * d[:, :, None, None]
        return upfirdn2d(y, blur_kernel, pad=blur_pad)
    pad = k // 2 if padding is None else padding
    w_t = w.permute(2, 3, 1, 0).contiguous()
    if fused:
        oh, ow = x.shape[2] + 2 * pad - k + 1, x.shape[3] + 2 * pad - k + 1
        return be.conv2d(x.contiguous(), w_t, s.contiguous(), None if d is None else d.contiguous(),
                         ConvGeom(k, k, 1, 1, pad, pad, oh, ow))
    y = conv2d_gradfix.conv2d_t(x * s[:, :, None, None], w_t, stride=1, padding=pad)
    if d is not None:
        y = y * d[:, :, None, None]
    return y
