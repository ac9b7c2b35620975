"""Differentiable weight re-layouts on gc_weight_layout_f32 (one pass: scale + permute + optional tap mirror).

The convolution kernels take weights as ``w_t = [kh, kw, K, N]`` in correlation order; parameters are stored the way the
reference stores them (``[N, K, kh, kw]``, gan_model.py:139, or ``[1, N, K, kh, kw]``, gan_model.py:268).  Every re-layout
is a scaled permutation P, so its gradient is the re-layout with source and destination swapped (P^T = P^-1 up to the
scale) and the closure needed for R1 / path-length double-backward is immediate.
"""
from torch.autograd import Function

from . import _backend
from . import weight_cache


class _Relayout(Function):
    """spec = (taps, k, n, src_stride, dst_shape, dst_stride, flip); strides are for the logical axes (tap, k, n)."""

    # (storage address, version, spec, scale) -> the cached re-layout of the CURRENT cache generation: the ~100 hits per pass skip the cache's
    # own look-up.  The memo owns its tensors, so it is emptied whenever the generation moves (every optimiser step / EMA accumulate): entries of
    # an older generation could never hit again and would only keep the re-layouts alive that weight_cache._drop has just released.
    _memo = {}
    _memo_generation = [-1]

    @staticmethod
    def forward(ctx, src, spec, scale):
        taps, k, n, src_stride, dst_shape, dst_stride, flip = spec
        ctx.inverse = (taps, k, n, dst_stride, tuple(src.shape), src_stride, flip)
        ctx.scale = scale
        fast = (src.data_ptr(), src._version, spec, scale) if weight_cache.ENABLED else None
        if _Relayout._memo_generation[0] != weight_cache.generation[0]:
            _Relayout._memo.clear()
            _Relayout._memo_generation[0] = weight_cache.generation[0]
        hit = _Relayout._memo.get(fast) if fast is not None else None
        if hit is not None:
            weight_cache.stats['hit'] += 1
            return hit.detach()
        # once per (weight, optimiser step): the result is cached on the weight's version counter (weight_cache.py); .detach() = a
        # fresh alias for autograd to attach this node to
        key = ('layout', taps, k, n, tuple(src_stride), tuple(dst_shape), tuple(dst_stride), bool(flip), float(scale))
        out = weight_cache.derive(src, key, lambda: _backend.get().weight_layout(src.contiguous(), taps, k, n, src_stride, dst_shape, dst_stride, flip, scale),
                                  recipe=key)
        if fast is not None and weight_cache._derived.get(out.data_ptr()) is not None:
            # only what the cache itself holds (a parameter's or a cached tensor's form); derive() may have bumped the generation (a refill
            # after an invalidation): the entry then belongs to the new generation
            if _Relayout._memo_generation[0] != weight_cache.generation[0]:
                _Relayout._memo.clear()
                _Relayout._memo_generation[0] = weight_cache.generation[0]
            _Relayout._memo[fast] = out
        return out.detach()

    @staticmethod
    def backward(ctx, g):
        return _backend.call(_Relayout, g, ctx.inverse, ctx.scale), None, None


def kernel_layout(weight, scale=1.0, flip=False, in_major=False):
    """Parameter -> w_t [kh, kw, K, N] (times scale, taps mirrored if flip).

    weight is [N, K, kh, kw] (F.conv2d layout) or, with in_major, [K, N, kh, kw] (F.conv_transpose2d layout).
    """
    d0, d1, kh, kw = weight.shape
    taps = kh * kw
    if in_major:
        k, n = d0, d1
        src_stride = (1, n * taps, taps)
    else:
        n, k = d0, d1
        src_stride = (1, taps, k * taps)
    return _backend.call(_Relayout, weight, (taps, k, n, src_stride, (kh, kw, k, n), (k * n, n, 1), bool(flip)), float(scale))


def adjoint_layout(w_t):
    """[kh, kw, K, N] -> [kh, kw, N, K] with mirrored taps: the weights of d/dx of the convolution."""
    kh, kw, k, n = w_t.shape
    return _backend.call(_Relayout, w_t, (kh * kw, k, n, (k * n, n, 1), (kh, kw, n, k), (n * k, 1, k), True), 1.0)
