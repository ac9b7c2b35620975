"""Bilinear affine warp and reflect padding with gradients, on gc_affine_warp_bilinear_f32 / gc_reflect_pad_f32.

The image-space pieces of the ADA augmentation (non_leaking.py:288-371) next to its two FIR passes.  Both are linear maps of the
image: forward and adjoint are each other's derivatives, which closes them under differentiation of any order.
"""
from torch.autograd import Function

from . import _backend


class _AffineWarp(Function):
    """y[b, c, oy, ox] = bilinear sample of x[b, c] at (m0*ox + m1*oy + m2, m3*ox + m4*oy + m5), zeros outside; mat is [B, 6]."""

    @staticmethod
    def forward(ctx, x, mat, out_h, out_w, adjoint, in_h, in_w):
        ctx.save_for_backward(mat)
        ctx.cfg = (out_h, out_w, adjoint, in_h, in_w)
        ctx.set_materialize_grads(False)
        return _backend.get().affine_warp(x.contiguous(), mat, in_h, in_w, out_h, out_w, adjoint)

    @staticmethod
    def backward(ctx, g):
        if g is None or not ctx.needs_input_grad[0]:
            return (None,) * 7
        mat, = ctx.saved_tensors
        out_h, out_w, adjoint, in_h, in_w = ctx.cfg
        return _backend.call(_AffineWarp, g, mat, out_h, out_w, not adjoint, in_h, in_w), None, None, None, None, None, None


def affine_warp_bilinear(x, mat, out_h, out_w):
    """x [B, C, H, W], mat [B, 6] (device, float32): F.grid_sample(bilinear, zeros, align_corners=False) for an affine grid."""
    return _backend.call(_AffineWarp, x, mat.contiguous(), int(out_h), int(out_w), False, int(x.shape[2]), int(x.shape[3]))


class _ReflectPad(Function):
    @staticmethod
    def forward(ctx, x, pads, adjoint, in_hw):
        ctx.cfg = (pads, adjoint, in_hw)
        ctx.set_materialize_grads(False)
        return _backend.get().reflect_pad(x.contiguous(), pads, adjoint, in_hw)

    @staticmethod
    def backward(ctx, g):
        if g is None or not ctx.needs_input_grad[0]:
            return None, None, None, None
        pads, adjoint, in_hw = ctx.cfg
        return _backend.call(_ReflectPad, g, pads, not adjoint, in_hw), None, None, None


def reflect_pad(x, pads):
    """F.pad(x, (left, right, top, bottom), mode='reflect') for [B, C, H, W]; raises ValueError when a pad is not smaller than the image."""
    left, right, top, bottom = (int(p) for p in pads)
    h, w = int(x.shape[2]), int(x.shape[3])
    if min(left, right, top, bottom) < 0 or max(left, right) >= w or max(top, bottom) >= h:
        raise ValueError(f'reflect_pad: padding {(left, right, top, bottom)} must be non-negative and smaller than the image {h} x {w}')
    return _backend.call(_ReflectPad, x, (left, right, top, bottom), False, (h, w))
