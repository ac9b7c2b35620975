"""upfirdn2d with first- and second-order gradients, on the HIP kernel gc_upfirdn2d_f32.

Socket: ``upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0))`` -- signature and defaults of the
reference wrapper gan_model.py:45-50 (which calls upfirdn2d_native, pytorch_upfirdn2d.py:9-51).
"""
import torch
from torch.autograd import Function

import os

from . import _backend

_FUSE_ACT_BWD = os.environ.get('GANCONTROL_FUSE_ACT_BWD_BLUR', '1') != '0'     # dev knob: 0 = activation backward and Blur adjoint as two launches


def _out_size(n, k, up, down, p0, p1):
    return (n * up + p0 + p1 - k) // down + 1


def _dense_or_pitched(x):
    """Contiguous, except a row-pitched tensor (the output of a transposed convolution, _lib.row_pitch): the FIR tile kernel reads it in place."""
    from ... import _lib
    return x if (x.is_cuda and _lib.row_pitch(x)) else x.contiguous()


class _UpFirDn2d(Function):
    @staticmethod
    def forward(ctx, x, kernel, up, down, p0, p1):
        kh, kw = kernel.shape
        n, c, h, w = x.shape
        oh, ow = _out_size(h, kh, up, down, p0, p1), _out_size(w, kw, up, down, p0, p1)
        if oh < 1 or ow < 1:
            raise ValueError(f'upfirdn2d: empty output for input {h}x{w}, kernel {kh}x{kw}, up={up}, down={down}, pad=({p0},{p1})')
        ctx.save_for_backward(kernel)
        ctx.cfg = (up, down, p0, p1, h, w)
        ctx.set_materialize_grads(False)        # a gradient nobody asked for arrives as None, not as a tensor of zeros to push through the kernels
        return _backend.get().upfirdn2d(_dense_or_pitched(x), kernel, up, down, p0, p0, oh, ow, True)

    @staticmethod
    def backward(ctx, gy):
        kernel, = ctx.saved_tensors
        gx = _backend.call(_UpFirDn2dAdjoint, gy, kernel, ctx.cfg) if (ctx.needs_input_grad[0] and gy is not None) else None
        return gx, None, None, None, None, None


class _UpFirDn2dAdjoint(Function):
    """gx = the same kernel with un-flipped taps, up <-> down and pad0' = k - 1 - pad0, sized like the input."""

    @staticmethod
    def forward(ctx, gy, kernel, cfg):
        up, down, p0, p1, h, w = cfg
        kh, kw = kernel.shape
        ctx.save_for_backward(kernel)
        ctx.cfg = cfg
        ctx.set_materialize_grads(False)        # a gradient nobody asked for arrives as None, not as a tensor of zeros to push through the kernels
        return _backend.get().upfirdn2d(_dense_or_pitched(gy), kernel, down, up, kw - 1 - p0, kh - 1 - p0, h, w, False)

    @staticmethod
    def backward(ctx, ggx):
        kernel, = ctx.saved_tensors
        up, down, p0, p1, h, w = ctx.cfg
        # the adjoint of the adjoint is the forward operator
        ggy = _backend.call(_UpFirDn2d, ggx, kernel, up, down, p0, p1) if (ctx.needs_input_grad[0] and ggx is not None) else None
        return ggy, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0), _internal=False):
    """The reference's socket signature.  ``_internal`` (modules of this package whose consumer reads a row pitch) lets the odd-width
    output of a Blur come back as a row-pitched view; every other caller gets an ordinary dense tensor."""
    if _internal:
        return _backend.call(_UpFirDn2d, input, kernel, int(up), int(down), int(pad[0]), int(pad[1]))
    with _backend.pitched_outputs(False):
        return _backend.call(_UpFirDn2d, input, kernel, int(up), int(down), int(pad[0]), int(pad[1]))


class _BlurOfActivation(Function):
    """y = upfirdn2d(a, kernel, pad=(p0, p1)) where ``a`` is the output of a fused bias + leaky-ReLU layer built with
    ``grad_premasked=True`` (conv2d_gradfix.conv2d_bias_act): this op's backward hands that layer the gradient w.r.t. its
    PRE-activation -- Blur adjoint and activation backward in ONE pass over the gradient (gc_upfirdn2d_mask_f32) instead of a FIR
    launch followed by an elementwise launch (ResBlock: conv1 -> FusedLeakyReLU -> Blur -> conv2, gan_model.py:893-922).

    The two ends belong together: ``a`` must have no other consumer (its producer skips its own mask multiplication)."""

    @staticmethod
    def forward(ctx, a, kernel, p0, p1, slope, gain):
        kh, kw = kernel.shape
        n, c, h, w = a.shape
        oh, ow = _out_size(h, kh, 1, 1, p0, p1), _out_size(w, kw, 1, 1, p0, p1)
        ctx.save_for_backward(kernel, a)
        ctx.cfg, ctx.act = (1, 1, p0, p1, h, w), (slope, gain)
        ctx.set_materialize_grads(False)
        return _backend.get().upfirdn2d(_dense_or_pitched(a), kernel, 1, 1, p0, p0, oh, ow, True)

    @staticmethod
    def backward(ctx, gy):
        kernel, a = ctx.saved_tensors
        if gy is None or not ctx.needs_input_grad[0]:
            return None, None, None, None, None, None
        return _backend.call(_BlurAdjointMasked, gy, kernel, ctx.cfg, a, *ctx.act), None, None, None, None, None


class _BlurAdjointMasked(Function):
    """g_pre = upfirdn2d_adjoint(gy) * (a > 0 ? gain : gain * slope); linear in gy."""

    @staticmethod
    def forward(ctx, gy, kernel, cfg, a, slope, gain):
        up, down, p0, p1, h, w = cfg
        kh, kw = kernel.shape
        ctx.save_for_backward(kernel, a)
        ctx.cfg, ctx.act = cfg, (slope, gain)
        ctx.set_materialize_grads(False)
        be = _backend.get()
        fused = getattr(be, 'upfirdn2d_mask', None)
        if (fused is not None and not _backend.strict_zeros() and a.is_contiguous()
                and be.upfirdn2d_act_supported(kernel, 1, 1, h, w, a.shape[0] * a.shape[1])):
            return fused(_dense_or_pitched(gy), kernel, kw - 1 - p0, kh - 1 - p0, h, w, False, a, slope, gain)
        g = be.upfirdn2d(_dense_or_pitched(gy), kernel, down, up, kw - 1 - p0, kh - 1 - p0, h, w, False)
        return be.bias_act_bwd(g.contiguous(), a.contiguous(), slope, gain)

    @staticmethod
    def backward(ctx, gg):
        from .fused_act import _BiasActGrad
        kernel, a = ctx.saved_tensors
        if gg is None:
            return None, None, None, None, None, None
        up, down, p0, p1, h, w = ctx.cfg
        slope, gain = ctx.act
        ggy = _backend.call(_UpFirDn2d, _backend.call(_BiasActGrad, gg, a, slope, gain), kernel, up, down, p0, p1) if ctx.needs_input_grad[0] else None
        # d/da of the mask is zero almost everywhere (zeros only for the trainer's dry run, as in _BiasActGrad)
        ga = torch.zeros_like(a) if (ctx.needs_input_grad[3] and _backend.strict_zeros()) else None
        return ggy, None, None, ga, None, None


def blur_of_activation(a, kernel, pad, negative_slope, scale):
    """upfirdn2d(a, kernel, pad=pad) for an ``a`` produced with ``grad_premasked=True`` (see _BlurOfActivation)."""
    return _backend.call(_BlurOfActivation, a, kernel, int(pad[0]), int(pad[1]), float(negative_slope), float(scale))


class _UpFirDn2dAct(Function):
    """out = gain * lrelu(FIR(x) + noise_w * noise + bias) in one launch (gc_upfirdn2d_act_f32): Blur -> NoiseInjection ->
    FusedLeakyReLU of an up-sampling StyledConv.  Backward = FusedLeakyReLU's (mask from the output, bias / noise-strength
    sums in the same pass) followed by the FIR adjoint; every piece is a differentiable Function."""

    @staticmethod
    def forward(ctx, x, kernel, p0, p1, bias, noise, noise_w, slope, gain):
        kh, kw = kernel.shape
        n, c, h, w = x.shape
        oh, ow = _out_size(h, kh, 1, 1, p0, p1), _out_size(w, kw, 1, 1, p0, p1)
        out = _backend.get().upfirdn2d_act(_dense_or_pitched(x), kernel, p0, p0, oh, ow, True, bias, None if noise is None else noise.contiguous(), noise_w, slope, gain)
        ctx.cfg, ctx.act = (1, 1, p0, p1, h, w), (slope, gain)
        ctx.has_bias, ctx.has_noise = bias is not None, noise is not None
        empty = x.new_empty(0)
        ctx.save_for_backward(kernel, out, noise if noise is not None else empty, noise_w if noise_w is not None else empty)
        ctx.set_materialize_grads(False)        # a gradient nobody asked for arrives as None, not as a tensor of zeros to push through the kernels
        return out

    @staticmethod
    def backward(ctx, gy):
        from .fused_act import _BiasActGrad, _BiasActGradReduce
        kernel, out, noise, noise_w = ctx.saved_tensors
        slope, gain = ctx.act
        need = ctx.needs_input_grad
        gx = gb = gnw = None
        params = _backend.want_param_grads()
        want_b, want_nw = ctx.has_bias and need[4] and params, ctx.has_noise and need[6] and params
        if gy is None or not (need[0] or want_b or want_nw):
            return (None,) * 9
        be = _backend.get()
        up, down, p0, p1, h, w = ctx.cfg
        kh, kw = kernel.shape
        fused = getattr(be, 'upfirdn2d_actbwd', None)
        if (fused is not None and _FUSE_ACT_BWD and need[0] and not torch.is_grad_enabled() and not _backend.strict_zeros()
                and gy.is_contiguous() and out.is_contiguous() and 0 <= p0 <= min(kh, kw) - 1
                and be.upfirdn2d_act_supported(kernel, 1, 1, h, w, gy.shape[0] * gy.shape[1])):
            # plain (not differentiated further) backward: activation backward, both reductions and the Blur adjoint in ONE pass over gy --
            # the pre-activation gradient never travels to HBM (gc_upfirdn2d_actbwd_f32).  Orders above one take the Functions below.
            gx, psum, pdot = fused(gy, out, noise.contiguous() if want_nw else None, kernel, kw - 1 - p0, kh - 1 - p0, h, w, False, slope, gain)
            if want_b:
                gb = psum.sum((0, 2))
            if want_nw:
                gnw = pdot.sum().reshape(noise_w.shape)
            return gx, None, None, None, gb, None, gnw, None, None
        if want_b or want_nw:
            g_pre, psum, pdot = _backend.call(_BiasActGradReduce, gy, out, noise if want_nw else None, slope, gain)[:3]
            if want_b:
                gb = psum.sum((0, 2))
            if want_nw:
                gnw = pdot.sum().reshape(noise_w.shape)
        else:
            g_pre = _backend.call(_BiasActGrad, gy, out, slope, gain)
        if need[0]:
            gx = _backend.call(_UpFirDn2dAdjoint, g_pre, kernel, ctx.cfg)
        return gx, None, None, None, gb, None, gnw, None, None


def upfirdn2d_bias_act(input, kernel, pad, bias, noise=None, noise_weight=None, negative_slope=0.2, scale=2 ** 0.5):
    """scale * lrelu(upfirdn2d(input, kernel, pad=pad) + noise_weight * noise + bias): one launch where the tile kernel applies,
    otherwise upfirdn2d followed by fused_noise_bias_act."""
    from .fused_act import fused_noise_bias_act
    p0, p1 = int(pad[0]), int(pad[1])
    kh, kw = kernel.shape
    oh, ow = _out_size(input.shape[2], kh, 1, 1, p0, p1), _out_size(input.shape[3], kw, 1, 1, p0, p1)
    if oh >= 1 and ow >= 1 and _backend.get().upfirdn2d_act_supported(kernel, 1, 1, oh, ow, input.shape[0] * input.shape[1]):
        if (noise is None) != (noise_weight is None):
            raise ValueError('noise and noise_weight go together')
        if noise is not None and (noise.shape[0] != input.shape[0] or noise.numel() != input.shape[0] * oh * ow):
            raise ValueError(f'noise shape {tuple(noise.shape)} does not match the output [{input.shape[0]}, {input.shape[1]}, {oh}, {ow}]')
        return _backend.call(_UpFirDn2dAct, input, kernel, p0, p1, None if bias is None else bias.reshape(-1).contiguous(), noise,
                                   None if noise_weight is None else noise_weight.reshape(-1).contiguous(), float(negative_slope), float(scale))
    return fused_noise_bias_act(upfirdn2d(input, kernel, pad=pad), bias, noise, noise_weight, negative_slope, scale)
