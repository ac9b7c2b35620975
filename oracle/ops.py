"""Operator-level oracle (plain PyTorch, any device/dtype, differentiable to any order).

Every function restates one reference operator and cites it.  The style is
functional on purpose: parameters come in as tensors, so the same code checks
the reference modules (make_golden.py) and the HIP product (tests/).

Test infrastructure only -- see oracle/__init__.py.
"""
import math

import torch
import torch.nn.functional as F

SQRT2 = math.sqrt(2.0)


def fir_kernel(taps, gain=1.0):
    """Normalised 2-D FIR from 1-D taps (outer product / sum), times ``gain``.

    Reference: make_kernel, gan_model.py:60-68; gain = factor**2 in
    Upsample (gan_model.py:76) and Blur(upsample_factor) (gan_model.py:119-120).
    """
    k = torch.as_tensor(taps, dtype=torch.float32)
    if k.ndim == 1:
        k = torch.outer(k, k)
    return k / k.sum() * gain


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """Upsample (zero-stuff) -> pad/crop -> FIR with flipped kernel -> decimate.

    Reference: upfirdn2d wrapper gan_model.py:45-50 -> upfirdn2d_native
    pytorch_upfirdn2d.py:9-51 (same pad on both axes: pad=(p0, p1)).

      out[n,c,oy,ox] = sum_{a,b} K[kh-1-a, kw-1-b] * U[oy*down + a - p0, ox*down + b - p0]
      U[y,x] = x[n,c,y/up,x/up] when up | y, up | x and inside, else 0.
    """
    n, c, h, w = x.shape
    kh, kw = kernel.shape
    p0, p1 = int(pad[0]), int(pad[1])
    # zero-stuffed plane of size (h*up, w*up): samples sit at multiples of `up`
    u = x.new_zeros(n * c, 1, h * up, w * up)
    u[:, :, ::up, ::up] = x.reshape(n * c, 1, h, w)
    # F.pad with negative amounts crops, which is what the reference's slice does
    u = F.pad(u, [p0, p1, p0, p1])
    taps = torch.flip(kernel, [0, 1]).to(x.dtype).reshape(1, 1, kh, kw)
    full = F.conv2d(u, taps)
    out = full[:, :, ::down, ::down]
    return out.reshape(n, c, out.shape[2], out.shape[3])


def fused_leaky_relu(x, bias, negative_slope=0.2, scale=SQRT2):
    """scale * leaky_relu(x + bias[c]) with bias broadcast over dim 1.

    Reference: FusedLeakyReLU.forward gan_model.py:32-35 and
    fused_leaky_relu gan_model.py:39-41.
    """
    shape = [1, -1] + [1] * (x.ndim - 2)
    return F.leaky_relu(x + bias.reshape(shape), negative_slope) * scale


def pixel_norm(x):
    """Reference: PixelNorm gan_model.py:52-57."""
    return x * torch.rsqrt(x.pow(2).mean(dim=1, keepdim=True) + 1e-8)


def equal_linear(x, weight, bias, lr_mul=1.0, activation=False):
    """Reference: EqualLinear.forward gan_model.py:189-197 (weight stored /lr_mul)."""
    scale = lr_mul / math.sqrt(weight.shape[1])
    if activation:
        return fused_leaky_relu(F.linear(x, weight * scale), bias * lr_mul)
    return F.linear(x, weight * scale, None if bias is None else bias * lr_mul)


def equal_conv2d(x, weight, bias=None, stride=1, padding=0):
    """Reference: EqualConv2d.forward gan_model.py:152-162."""
    oc, ic, kh, kw = weight.shape
    return F.conv2d(x, weight * (1.0 / math.sqrt(ic * kh * kw)), bias, stride=stride, padding=padding)


def modulated_conv2d(x, style, weight, mod_weight, mod_bias, demodulate=True, upsample=False,
                     blur_kernel=(1, 3, 3, 1)):
    """Weight-(de)modulated grouped conv, per-sample weights materialised as the reference does.

    Reference: ModulatedConv2d.forward gan_model.py:281-331 (plain branch :325-329,
    conv_transpose upsample branch :295-307); weight is [1, OC, IC, k, k];
    modulation = EqualLinear(style_dim, IC, bias_init=1) gan_model.py:271.
    """
    b, ic, h, w = x.shape
    _, oc, _, k, _ = weight.shape
    s = equal_linear(style, mod_weight, mod_bias)                       # [B, IC]
    wb = weight * (1.0 / math.sqrt(ic * k * k)) * s.reshape(b, 1, ic, 1, 1)
    if demodulate:
        wb = wb * torch.rsqrt(wb.pow(2).sum([2, 3, 4], keepdim=True) + 1e-8)
    if upsample:
        wt = wb.transpose(1, 2).reshape(b * ic, oc, k, k)
        y = F.conv_transpose2d(x.reshape(1, b * ic, h, w), wt, stride=2, padding=0, groups=b)
        y = y.reshape(b, oc, y.shape[2], y.shape[3])
        p = (len(blur_kernel) - 2) - (k - 1)
        y = upfirdn2d(y, fir_kernel(blur_kernel, 4.0).to(y), pad=((p + 1) // 2 + 1, p // 2 + 1))
        return y
    y = F.conv2d(x.reshape(1, b * ic, h, w), wb.reshape(b * oc, ic, k, k), padding=k // 2, groups=b)
    return y.reshape(b, oc, y.shape[2], y.shape[3])


def minibatch_stddev(x, group_size=4):
    """Append the mini-batch standard-deviation channel.

    Reference: Discriminator._forward_split gan_model.py:1003-1012 (stddev_feat = 1).
    Members of a group are strided: i, i + B/G, i + 2B/G, ...
    """
    b, c, h, w = x.shape
    g = min(b, group_size)
    y = x.reshape(g, b // g, 1, c, h, w)
    y = torch.sqrt(y.var(0, unbiased=False) + 1e-8)
    y = y.mean([2, 3, 4], keepdim=True).squeeze(2)
    return torch.cat([x, y.repeat(g, 1, h, w)], 1)
