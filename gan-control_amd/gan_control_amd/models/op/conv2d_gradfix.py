"""conv2d / conv_transpose2d with arbitrary-order gradients on the MFMA kernels
gc_conv2d_f32 / gc_conv2d_wgrad_f32.

The reference calls plain ATen (F.conv2d gan_model.py:154,327; F.conv_transpose2d gan_model.py:304)
and relies on ATen being infinitely differentiable for R1 (generator_trainer.py:713-719) and
path-length regularisation (gan_model.py:803-811).  Here the same closure is obtained with two
autograd Functions that are each other's derivatives:

  _GConv(x, w_t)   y[b,n,o]    = sum_{t,k} w_t[t,k,n] * U_up(x)[b,k, o*down + t - pad]
  _WGrad(x, gy)    g[t,k,n]    = sum_{b,o} x[b,k, o*down + t - pad] * gy[b,n,o]

d_GConv/dx is a _GConv with (up <-> down, flipped taps, swapped channel axes); d_GConv/dw is a
_WGrad; both derivatives of _WGrad are _GConv's.  Weights travel as w_t = [kh, kw, K, N]
(correlation order, N contiguous); the layout changes are single fused passes (weight_layout.py).
"""
import os

import torch
from torch.autograd import Function

from . import _backend
from .upfirdn2d import _dense_or_pitched
from ._backend import ConvGeom
from .weight_layout import adjoint_layout, kernel_layout


_FUSE_PW_ACT = os.environ.get('GANCONTROL_FUSE_PW_ACT', '1') != '0'      # dev knob: 0 = FromRGB's activation backward as a launch of its own


def _adjoint_geom(g, in_h, in_w):
    """Geometry of d/dx: swaps up/down, mirrors the pad, and produces the input extent."""
    return ConvGeom(g.kh, g.kw, g.down, g.up, g.kh - 1 - g.pad_y, g.kw - 1 - g.pad_x, in_h, in_w)


def _adjoint_weight(w_t):
    """[kh,kw,K,N] -> [kh,kw,N,K] with flipped taps (one fused pass)."""
    return adjoint_layout(w_t)


class _GConv(Function):
    """y = gconv(x, w_t) [+ residual].  The residual rides in the convolution's epilogue (gc_conv_epilogue.residual): the
    `out + skip` of a ResBlock, and on the backward side the gradient coming from the other consumer of a forked tensor."""

    @staticmethod
    def forward(ctx, x, w_t, geom, residual=None):
        ctx.geom = geom
        ctx.in_hw = (x.shape[2], x.shape[3])
        ctx.save_for_backward(x, w_t)
        ctx.set_materialize_grads(False)
        ep = None if residual is None else (None, None, None, 1.0, 1.0, False, residual.contiguous())
        return _backend.get().conv2d(_dense_or_pitched(x), w_t.contiguous(), None, None, geom, epilogue=ep)

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return (None,) * len(ctx.needs_input_grad)
        x, w_t = ctx.saved_tensors
        g = ctx.geom
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _backend.call(_GConv, gy, _adjoint_weight(w_t), _adjoint_geom(g, *ctx.in_hw))
        if ctx.needs_input_grad[1] and _backend.want_param_grads():
            gw = _weight_grad(x, gy, g)
        want_res = len(ctx.needs_input_grad) > 3 and ctx.needs_input_grad[3]
        return (gx, gw, None, gy if want_res else None)[:len(ctx.needs_input_grad)]


def _weight_grad(x, gy, g):
    if g.up == 1:
        return _backend.call(_WGrad, x, gy, g)
    # transposed conv: correlate gy (as the "input", decimated by `up`) with x (as the "output gradient")
    swapped = ConvGeom(g.kh, g.kw, 1, g.up, g.kh - 1 - g.pad_y, g.kw - 1 - g.pad_x, x.shape[2], x.shape[3])
    return adjoint_layout(_backend.call(_WGrad, gy, x, swapped))


class _WGrad(Function):
    @staticmethod
    def forward(ctx, x, gy, geom):
        ctx.geom = geom
        ctx.in_hw = (x.shape[2], x.shape[3])
        ctx.save_for_backward(x, gy)
        ctx.set_materialize_grads(False)
        return _backend.get().conv2d_wgrad(_dense_or_pitched(x), _dense_or_pitched(gy), None, None, geom)

    @staticmethod
    def backward(ctx, ggw):
        if ggw is None:
            return None, None, None
        x, gy = ctx.saved_tensors
        g = ctx.geom
        gx = ggy = None
        if ctx.needs_input_grad[0]:
            gx = _backend.call(_GConv, gy, _adjoint_weight(ggw), _adjoint_geom(g, *ctx.in_hw))
        if ctx.needs_input_grad[1]:
            ggy = _backend.call(_GConv, x, ggw.contiguous(), g)
        return gx, ggy, None


class _GConvAct(Function):
    """out = gain * lrelu(gconv(x, w_t) + bias): the activation runs in the convolution's epilogue (gc_conv_epilogue), so
    the pre-activation tensor is never written.  Backward = FusedLeakyReLU's (mask from the sign of the OUTPUT, bias
    gradient reduced in the same pass) followed by _GConv's; every piece is a differentiable Function, so R1 closes.

    fork=True additionally returns x itself as a second output for the OTHER consumer of x (the skip branch of a ResBlock).
    That consumer's gradient then arrives here as the second output-gradient and is added inside the input-gradient
    convolution (residual epilogue) instead of by autograd's separate elementwise add over the largest tensors of D.
    """

    @staticmethod
    def forward(ctx, x, w_t, bias, geom, slope, gain, fork=False, premasked=False):
        out = _backend.get().conv2d(_dense_or_pitched(x), w_t.contiguous(), None, None, geom, epilogue=(bias, None, None, slope, gain, True))
        ctx.geom, ctx.cfg = geom, (slope, gain)
        ctx.premasked = bool(premasked)
        ctx.in_hw = (x.shape[2], x.shape[3])
        ctx.save_for_backward(x, w_t, out)
        ctx.set_materialize_grads(False)
        return (out, x.view_as(x)) if fork else out

    @staticmethod
    def backward(ctx, gy, gfork=None):
        from .fused_act import _BiasActGrad, _BiasActGradReduce
        x, w_t, out = ctx.saved_tensors
        g = ctx.geom
        slope, gain = ctx.cfg
        gx = gw = gb = None
        if not any(ctx.needs_input_grad[:3]):
            return None, None, None, None, None, None, None, None
        if gy is None:                      # only the forked copy was used downstream
            return (gfork if ctx.needs_input_grad[0] else None), None, None, None, None, None, None, None
        params = _backend.want_param_grads()
        be = _backend.get()
        if (_FUSE_PW_ACT and not ctx.premasked and gfork is None and g.kh == 1 and g.kw == 1 and g.up == 1 and g.down == 1 and g.pad_y == 0
                and not torch.is_grad_enabled() and not _backend.strict_zeros() and getattr(be, 'pw_act_wgrad', None) is not None
                and gy.is_contiguous() and out.is_contiguous() and x.is_contiguous() and be.pw_act_supported(x, gy)):
            # D's FromRGB layer (3 -> C, 1x1, on the largest planes of the network): its activation backward rides in the loads of the
            # weight-gradient / input-gradient kernels (gc_pw_act_wgrad_f32 / gc_pw_act_dgrad_f32) instead of being a pass of its own
            if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and params:
                gw_, gb_ = be.pw_act_wgrad(x, gy, out, slope, gain)
                gw = gw_ if ctx.needs_input_grad[1] else None
                gb = gb_ if ctx.needs_input_grad[2] else None
            if ctx.needs_input_grad[0]:
                gx = be.pw_act_dgrad(gy, out, _adjoint_weight(w_t).contiguous(), slope, gain)
            return gx, gw, gb, None, None, None, None, None
        if ctx.premasked:
            # the only consumer of `out` (upfirdn2d.blur_of_activation) has applied this activation's mask already: gy IS the gradient
            # of the pre-activation
            from .fused_act import _channel_sum
            g_pre = gy
            if ctx.needs_input_grad[2] and params:
                gb = _channel_sum(gy)
        elif ctx.needs_input_grad[2] and params:
            g_pre, psum = _backend.call(_BiasActGradReduce, gy, out, None, slope, gain)[:2]
            gb = psum.sum((0, 2))
        else:
            g_pre = _backend.call(_BiasActGrad, gy, out, slope, gain)
        if ctx.needs_input_grad[0]:
            gx = _backend.call(_GConv, g_pre, _adjoint_weight(w_t), _adjoint_geom(g, *ctx.in_hw), gfork)
        if ctx.needs_input_grad[1] and params:
            gw = _weight_grad(x, g_pre, g)
        return gx, gw, gb, None, None, None, None, None


def _pair(v):
    return (int(v), int(v)) if not isinstance(v, (tuple, list)) else (int(v[0]), int(v[1]))


def _check(input, weight, stride, padding, dilation, groups):
    if dilation not in (1, (1, 1)) or groups != 1:
        raise ValueError('conv2d_gradfix: only dilation=1, groups=1 are implemented on the HIP path')
    s, p = _pair(stride), _pair(padding)
    if s[0] != s[1] or p[0] != p[1]:
        raise ValueError('conv2d_gradfix: stride and padding must be square')
    if input.ndim != 4 or weight.ndim != 4:
        raise ValueError('conv2d_gradfix: 4-D input and weight expected')
    return s[0], p[0]


def conv2d_t(x, w_t, stride=1, padding=0, residual=None):
    """conv2d with the weight already in [kh, kw, IC, OC] layout (+ residual, added in the kernel's epilogue)."""
    kh, kw = w_t.shape[0], w_t.shape[1]
    oh = (x.shape[2] + 2 * padding - kh) // stride + 1
    ow = (x.shape[3] + 2 * padding - kw) // stride + 1
    return _backend.call(_GConv, x, w_t, ConvGeom(kh, kw, 1, stride, padding, padding, oh, ow), residual)


def conv_transpose2d_t(x, w_t, stride=1, padding=0):
    """conv_transpose2d with the weight already in CORRELATION form [kh, kw, IC, OC] (taps flipped)."""
    kh, kw = w_t.shape[0], w_t.shape[1]
    oh = (x.shape[2] - 1) * stride - 2 * padding + kh
    ow = (x.shape[3] - 1) * stride - 2 * padding + kw
    return _backend.call(_GConv, x, w_t, ConvGeom(kh, kw, stride, 1, kh - 1 - padding, kw - 1 - padding, oh, ow))


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, weight_scale=1.0, residual=None):
    """Same call signature as torch.nn.functional.conv2d (weight [OC, IC, kh, kw]); weight_scale (an extension) folds
    the equalised-learning-rate factor into the layout pass instead of a separate ``weight * scale``."""
    s, p = _check(input, weight, stride, padding, dilation, groups)
    if weight.shape[1] != input.shape[1]:
        raise ValueError(f'conv2d: weight expects {weight.shape[1]} input channels, got {input.shape[1]}')
    if residual is not None and bias is not None:
        raise NotImplementedError('conv2d: residual together with a plain bias is not built')
    y = conv2d_t(input, kernel_layout(weight, weight_scale), s, p, residual)
    return y if bias is None else y + bias.reshape(1, -1, 1, 1)


def conv2d_bias_act(input, weight, bias, stride=1, padding=0, weight_scale=1.0, negative_slope=0.2, scale=2 ** 0.5, fork=False, grad_premasked=False):
    """scale * leaky_relu(conv2d(input, weight * weight_scale) + bias): EqualConv2d -> FusedLeakyReLU (ConvLayer,
    gan_model.py:844-890) as ONE kernel launch.  grad_premasked=True: the result's ONLY consumer is upfirdn2d.blur_of_activation, whose
    backward returns the gradient of the pre-activation (this layer then skips its own activation-backward pass)."""
    s, p = _check(input, weight, stride, padding, 1, 1)
    if weight.shape[1] != input.shape[1]:
        raise ValueError(f'conv2d: weight expects {weight.shape[1]} input channels, got {input.shape[1]}')
    if bias.numel() != weight.shape[0]:
        raise ValueError(f'conv2d_bias_act: bias has {bias.numel()} elements, weight has {weight.shape[0]} output channels')
    kh, kw = weight.shape[2], weight.shape[3]
    oh = (input.shape[2] + 2 * p - kh) // s + 1
    ow = (input.shape[3] + 2 * p - kw) // s + 1
    return _backend.call(_GConvAct, input, kernel_layout(weight, weight_scale), bias.reshape(-1).contiguous(),
                           ConvGeom(kh, kw, 1, s, p, p, oh, ow), float(negative_slope), float(scale), bool(fork), bool(grad_premasked))


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1, weight_scale=1.0):
    """Same call signature as torch.nn.functional.conv_transpose2d (weight [IC, OC, kh, kw]) plus weight_scale."""
    s, p = _check(input, weight, stride, padding, dilation, groups)
    if output_padding not in (0, (0, 0)):
        raise ValueError('conv_transpose2d: output_padding is not implemented on the HIP path')
    if weight.shape[0] != input.shape[1]:
        raise ValueError(f'conv_transpose2d: weight expects {weight.shape[0]} input channels, got {input.shape[1]}')
    with _backend.pitched_outputs(False):         # a public result: dense rows (the row-pitched layout stays inside this package)
        y = conv_transpose2d_t(input, kernel_layout(weight, weight_scale, flip=True, in_major=True), s, p)
    return y if bias is None else y + bias.reshape(1, -1, 1, 1)
