"""Fused (noise +) bias + leaky-ReLU * gain with first- and second-order gradients.

Socket: ``FusedLeakyReLU(channel, negative_slope=0.2, scale=2 ** 0.5)`` and
``fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5)`` -- names, signatures and
defaults of gan_model.py:25-41.  ``fused_noise_bias_act`` additionally folds the NoiseInjection
add that precedes every activation in StyledConv (gan_model.py:340-345, 402-408) into the same pass.
"""
import torch
from torch import nn
from torch.autograd import Function

from . import _backend


def _channel_sum(t):
    return _backend.call(_ChannelSum, t)


class _ChannelSum(Function):
    """[B, C, *] -> [C] on gc_channel_sum_f32 (deterministic); differentiable (broadcast back)."""

    @staticmethod
    def forward(ctx, t):
        ctx.shape = t.shape
        ctx.set_materialize_grads(False)
        return _backend.get().channel_sum(t.contiguous())

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None
        shape = ctx.shape
        return g.reshape([1, -1] + [1] * (len(shape) - 2)).expand(shape)


class _BiasAct(Function):
    @staticmethod
    def forward(ctx, x, bias, noise, noise_w, slope, gain):
        y = _backend.get().bias_act(x.contiguous(), bias, None if noise is None else noise.contiguous(), noise_w, slope, gain)
        ctx.cfg = (slope, gain)
        ctx.has_bias, ctx.has_noise = bias is not None, noise is not None
        ctx.save_for_backward(y, noise if noise is not None else y.new_empty(0), noise_w if noise_w is not None else y.new_empty(0))
        # (default gradient materialisation on purpose: the reference's leaky-ReLU yields ZERO second-order gradients through its mask,
        # and this public op keeps that None-vs-zero structure; the internal Functions below pass None on instead of zeros)
        return y

    @staticmethod
    def backward(ctx, gy):
        y, noise, noise_w = ctx.saved_tensors
        slope, gain = ctx.cfg
        gx = gb = gn = gnw = None
        if not any(ctx.needs_input_grad[:4]):
            return None, None, None, None, None, None
        params = _backend.want_param_grads()
        want_b = ctx.has_bias and ctx.needs_input_grad[1] and params
        want_n = ctx.has_noise and ctx.needs_input_grad[2]
        want_nw = ctx.has_noise and ctx.needs_input_grad[3] and params
        if (want_b or want_nw) and not want_n:
            # one pass: activation gradient + per-plane partial sums for the bias / noise-strength gradients
            gx, psum, pdot = _backend.call(_BiasActGradReduce, gy, y, noise if want_nw else None, slope, gain)[:3]
            if want_b:
                gb = psum.sum((0, 2))
            if want_nw:
                gnw = pdot.sum().reshape(noise_w.shape)
        else:
            gx = _backend.call(_BiasActGrad, gy, y, slope, gain)
            if want_b:
                gb = _channel_sum(gx)
            if want_n or want_nw:
                per_px = gx.sum(1, keepdim=True)                   # [B, 1, *]
                if want_n:
                    gn = per_px * noise_w
                if want_nw:
                    gnw = (per_px * noise).sum().reshape(noise_w.shape)
        if not ctx.needs_input_grad[0]:
            gx = None
        return gx, gb, gn, gnw, None, None


REDUCE_CHUNK = 16384      # elements of one plane summed per workgroup (channel_sum_plan in csrc/bias_act.hip)


def _spread(partial, like):
    """[B, C, chunks] cotangent of a per-chunk sum -> one value per element of `like`."""
    b, c = like.shape[0], like.shape[1]
    inner = like.numel() // (b * c)
    return partial.repeat_interleave(REDUCE_CHUNK, dim=2)[:, :, :inner].reshape(like.shape)


class _BiasActGradReduce(Function):
    """(gx, psum, pdot, pself) = gc_bias_act_bwd_reduce_self_f32; every output is linear in gy.

    pself (only with self_dot = True) = chunk sums of gx * x_pre, x_pre = the activation's input rebuilt from its output:
    x_pre = lrelu^-1(y / gain) - bias - noise_w * noise.  It makes the out_scale gradient of a convolution with a fused
    activation epilogue without the pre-activation tensor ever existing (see modulated_conv._ModConvAct).
    """

    @staticmethod
    def forward(ctx, gy, y, noise, slope, gain, bias=None, noise_w=None, self_dot=False):
        ctx.cfg = (slope, gain)
        ctx.has_noise, ctx.self_dot, ctx.has_bias = noise is not None, bool(self_dot), bias is not None
        nz = None if noise is None else noise.contiguous()
        gx, psum, pdot, pself = _backend.get().bias_act_bwd_reduce(gy.contiguous(), y, nz, slope, gain,
                                                                    self_dot=(bias, noise_w if noise is not None else None) if self_dot else None)
        empty = y.new_empty(0)
        ctx.save_for_backward(y, noise if noise is not None else empty, bias if (self_dot and bias is not None) else empty,
                              noise_w if (self_dot and noise is not None) else empty, gx if self_dot else empty)
        dead = []
        if pdot is None:
            pdot = psum.new_empty(0)
            dead.append(pdot)
        if pself is None:
            pself = psum.new_empty(0)
            dead.append(pself)
        if dead:
            ctx.mark_non_differentiable(*dead)
        ctx.set_materialize_grads(False)        # cotangents of outputs nobody used arrive as None: the adjoint pass skips their terms
        return gx, psum, pdot, pself

    @staticmethod
    def backward(ctx, ggx, gpsum, gpdot, gpself):
        if ggx is None and gpsum is None and gpdot is None and gpself is None:
            return (None,) * 8
        y, noise, bias, noise_w, gx = ctx.saved_tensors
        slope, gain = ctx.cfg
        b = y.shape[0]
        if not torch.is_grad_enabled() and slope != 0 and gain != 0:
            # the usual case (the regulariser's own backward): one fused pass; the ATen formulas below stay for higher orders
            def chunked(t):
                return None if (t is None or t.numel() == 0) else t.contiguous()
            cw = chunked(gpself) if ctx.self_dot else None
            g_gy, g_y, pgb, pgn = _backend.get().bias_act_bwd_reduce_adjoint(
                None if ggx is None else ggx.contiguous(), chunked(gpsum), chunked(gpdot) if ctx.has_noise else None, cw, y,
                gx if cw is not None else None, noise.contiguous() if ctx.has_noise else None, bias if (cw is not None and ctx.has_bias) else None,
                noise_w if (cw is not None and ctx.has_noise) else None, slope, gain, want_gyref=ctx.needs_input_grad[1])
            g_bias = g_nw = None
            if cw is not None and ctx.has_bias and ctx.needs_input_grad[5]:
                g_bias = -pgb.sum((0, 2))
            if cw is not None and ctx.has_noise and ctx.needs_input_grad[6]:
                g_nw = -pgn.sum().reshape(noise_w.shape)
            if g_y is None and ctx.needs_input_grad[1] and _backend.strict_zeros():
                g_y = torch.zeros_like(y)
            return (g_gy if ctx.needs_input_grad[0] else None), g_y, None, None, None, g_bias, g_nw, None
        noise4 = noise.reshape(b, 1, *y.shape[2:]) if ctx.has_noise else None
        total = torch.zeros_like(y) if ggx is None else ggx
        if gpsum is not None:
            total = total + _spread(gpsum, y)
        if ctx.has_noise and gpdot is not None and gpdot.numel() > 0:
            total = total + _spread(gpdot, y) * noise4
        g_y = g_bias = g_nw = None
        if ctx.self_dot and gpself is not None and gpself.numel() > 0:
            # x_pre materialised with ATen: this branch only runs in double-backward (path-length regularisation)
            w_self = _spread(gpself, y)
            x_pre = torch.where(y > 0, y / gain, y / (gain * slope))
            if ctx.has_bias:
                x_pre = x_pre - bias.reshape([1, -1] + [1] * (y.ndim - 2))
            if ctx.has_noise:
                x_pre = x_pre - noise_w * noise4
            total = total + w_self * x_pre
            wg = w_self * gx
            # d x_pre / d y = 1 / (gain * mask) and gx = gy * gain * mask  =>  wg * d x_pre / d y = w_self * gy
            if ctx.needs_input_grad[1]:
                g_y = torch.where(y > 0, wg / gain, wg / (gain * slope))
            if ctx.has_bias and ctx.needs_input_grad[5]:
                g_bias = -wg.sum([d for d in range(y.ndim) if d != 1])
            if ctx.has_noise and ctx.needs_input_grad[6]:
                g_nw = -(wg * noise4).sum().reshape(noise_w.shape)
        g_gy = _backend.call(_BiasActGrad, total, y, slope, gain) if ctx.needs_input_grad[0] else None
        if g_y is None and ctx.needs_input_grad[1] and _backend.strict_zeros():
            g_y = torch.zeros_like(y)
        return g_gy, g_y, None, None, None, g_bias, g_nw, None


class _BiasActGrad(Function):
    """gx = gy * (y > 0 ? gain : gain * slope); linear in gy, so it is its own adjoint."""

    @staticmethod
    def forward(ctx, gy, y, slope, gain):
        ctx.save_for_backward(y)
        ctx.cfg = (slope, gain)
        ctx.set_materialize_grads(False)
        return _backend.get().bias_act_bwd(gy.contiguous(), y, slope, gain)

    @staticmethod
    def backward(ctx, ggx):
        if ggx is None:
            return None, None, None, None
        y, = ctx.saved_tensors
        slope, gain = ctx.cfg
        ggy = _backend.call(_BiasActGrad, ggx, y, slope, gain) if ctx.needs_input_grad[0] else None
        # d/dy of the mask is zero almost everywhere: no gradient flows to the forward output (zeros only for the dry run)
        return ggy, (torch.zeros_like(y) if (ctx.needs_input_grad[1] and _backend.strict_zeros()) else None), None, None


def fused_noise_bias_act(input, bias=None, noise=None, noise_weight=None, negative_slope=0.2, scale=2 ** 0.5):
    """scale * lrelu(input + bias[c] + noise_weight * noise[b, 0]); bias broadcasts over dim 1."""
    if (noise is None) != (noise_weight is None):
        raise ValueError('noise and noise_weight go together')
    if bias is not None and bias.numel() != input.shape[1]:
        raise ValueError(f'bias has {bias.numel()} elements, input has {input.shape[1]} channels')
    if bias is not None:
        bias = bias.reshape(-1).contiguous()
    if noise is not None:
        if noise.shape[0] != input.shape[0] or noise.numel() * input.shape[1] != input.numel():
            raise ValueError(f'noise shape {tuple(noise.shape)} does not match input {tuple(input.shape)}')
        noise_weight = noise_weight.reshape(-1).contiguous()
    return _backend.call(_BiasAct, input, bias, noise, noise_weight, float(negative_slope), float(scale))


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    return fused_noise_bias_act(input, bias, None, None, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input, noise=None, noise_weight=None):
        return fused_noise_bias_act(input, self.bias, noise, noise_weight, self.negative_slope, self.scale)
