"""upfirdn2d with first- and second-order gradients, on the HIP kernel gc_upfirdn2d_f32.

Socket: ``upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0))`` -- signature and defaults of the
reference wrapper gan_model.py:45-50 (which calls upfirdn2d_native, pytorch_upfirdn2d.py:9-51).
"""
from torch.autograd import Function

from . import _backend


def _out_size(n, k, up, down, p0, p1):
    return (n * up + p0 + p1 - k) // down + 1


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
        return _backend.get().upfirdn2d(x.contiguous(), kernel, up, down, p0, p0, oh, ow, True)

    @staticmethod
    def backward(ctx, gy):
        kernel, = ctx.saved_tensors
        gx = _UpFirDn2dAdjoint.apply(gy, kernel, ctx.cfg) if ctx.needs_input_grad[0] else None
        return gx, None, None, None, None, None


class _UpFirDn2dAdjoint(Function):
    """gx = the same kernel with un-flipped taps, up <-> down and pad0' = k - 1 - pad0, sized like the input."""

    @staticmethod
    def forward(ctx, gy, kernel, cfg):
        up, down, p0, p1, h, w = cfg
        kh, kw = kernel.shape
        ctx.save_for_backward(kernel)
        ctx.cfg = cfg
        return _backend.get().upfirdn2d(gy.contiguous(), kernel, down, up, kw - 1 - p0, kh - 1 - p0, h, w, False)

    @staticmethod
    def backward(ctx, ggx):
        kernel, = ctx.saved_tensors
        up, down, p0, p1, h, w = ctx.cfg
        # the adjoint of the adjoint is the forward operator
        ggy = _UpFirDn2d.apply(ggx, kernel, up, down, p0, p1) if ctx.needs_input_grad[0] else None
        return ggy, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    return _UpFirDn2d.apply(input, kernel, int(up), int(down), int(pad[0]), int(pad[1]))
