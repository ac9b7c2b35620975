"""Weight-(de)modulated convolution without per-sample weights.

Reference: ModulatedConv2d.forward gan_model.py:281-331 materialises ``[B*OC, IC, k, k]`` weights
and runs a grouped conv with groups = B.  Algebraically the same result is

    y = d[b,oc] * conv(x * s[b,ic], W * scale),   d = rsqrt(scale^2 * sum_ic s^2 * sum_k W^2 + 1e-8)

i.e. ONE shared-weight convolution with a per-(b,ic) scale on the way in and a per-(b,oc) scale
on the way out (relative error vs the reference ~3e-7).  Both scales
are applied INSIDE the convolution kernels (while staging / in the epilogue), forward and backward:
no elementwise pass over the activations is spent on modulation.

Autograd closure (all orders): with  F(x, w, si, so) = so * gconv(si * x, w)

    dF/dx   = F(gy, adj(w), so, si)  with the adjoint geometry       -> _ModConv again
    dF/dw   = WG(x, gy, si, so)                                      -> _ModWGrad
    dF/dsi  = sum_hw x * (dF/dx) / si ;  dF/dso = sum_hw gy * y / so -> _PlaneDot (+ a [B,C] division)

and both derivatives of _ModWGrad are _ModConv's, so R1 / path-length double-backward close.

First-order shortcut (grad mode off, i.e. nothing will differentiate this backward pass again): dF/dsi and dF/dso are contractions of the
PER-SAMPLE weight gradient with the weight,  dF/dsi[b,k] = sum_{t,n} w[t,k,n] WG_b[t,k,n] / si[b,k]  (dF/dso alike over k), and the
weight-gradient kernel can hand out WG_b at no cost (its pixel splits regrouped by sample: gc_conv2d_wgrad_samples_*), so the two
full-plane products of _PlaneDot are not needed where the planes are large next to the weights (_samples_route).
"""
import math
import os

import torch
from torch.autograd import Function

from . import _backend
from ._backend import ConvGeom
from .conv2d_gradfix import _adjoint_geom, _adjoint_weight
from .weight_layout import kernel_layout
from .upfirdn2d import upfirdn2d, _dense_or_pitched


_WGRAD_SAMPLES = int(os.environ.get('GANCONTROL_WGRAD_SAMPLES', '1'))      # 0: off; 1: nodes of the forward pass; 2: also the input-gradient nodes met in a second backward
_SAMPLES_MIN_RATIO = float(os.environ.get('GANCONTROL_WGRAD_SAMPLES_RATIO', '2'))        # plane bytes the _PlaneDot's would read / bytes the per-sample route writes and re-reads


def _samples_route(x, gy, g, want_si, want_so, derived=False):
    """True where the scale gradients come from the per-sample weight gradient: first-order only (the route is not differentiable),
    a shape the backend has the per-sample form for, and planes large enough that skipping the plane products pays for writing and
    reading [B, taps, K, N] (at 512 channels and 64 x 64 pixels it does not).  At the default level 1 only nodes of the network's FORWARD
    pass take it: the scale gradient of an input-gradient node (`derived`, met in the path-length step's second backward) cancels exactly
    against its _PlaneDot partner's, which only the plane route -- the same tensor read on both sides -- preserves in split-bf16
    arithmetic (tools/samples_route_probe.py)."""
    if not _WGRAD_SAMPLES or torch.is_grad_enabled() or not (want_si or want_so) or (derived and _WGRAD_SAMPLES < 2):
        return False
    be = _backend.get()
    if not hasattr(be, 'conv2d_wgrad_samples_bytes'):
        return False
    saved = 2.0 * ((x.numel() if want_si else 0) + (gy.numel() if want_so else 0))
    extra = 3.0 * x.shape[0] * g.kh * g.kw * x.shape[1] * gy.shape[1]
    if saved < _SAMPLES_MIN_RATIO * extra:
        return False
    if g.up == 1:
        return be.conv2d_wgrad_samples_bytes(_dense_or_pitched(x), gy, g) > 0
    return be.conv2d_wgrad_samples_bytes(_dense_or_pitched(gy), x, _swapped_geom(g, x)) > 0


def _weight_and_scale_grads(x, gy, w_t, si, so, g, want_si, want_so):
    """(gw, gsi, gso) of F = so * gconv(si * x, w_t) from ONE weight-gradient launch (see the module docstring); gsi / gso None if not wanted."""
    be = _backend.get()
    cs = lambda t: None if t is None else t.contiguous()
    if g.up == 1:
        gw, dws = be.conv2d_wgrad_samples(_dense_or_pitched(x), _dense_or_pitched(gy), cs(si), cs(so), g)
        gsi, gso = be.wgrad_samples_contract(dws, w_t.contiguous(), cs(si), cs(so), want_si, want_so)
        return gw, gsi, gso
    # transposed convolution: the weight gradient is computed with the operands swapped, in the layout of the input-gradient weights
    # (_mod_weight_grad); the contraction runs in that layout too -- adj(w_t) is what the input-gradient convolution used a moment ago
    dw_adj, dws = be.conv2d_wgrad_samples(_dense_or_pitched(gy), _dense_or_pitched(x), cs(so), cs(si), _swapped_geom(g, x))
    gso, gsi = be.wgrad_samples_contract(dws, _adjoint_weight(w_t).contiguous(), cs(so), cs(si), want_so, want_si)
    return _adjoint_weight(dw_adj), gsi, gso


def _wsq_value(w):
    """sum over the taps of w^2, [OC, IC]: the backend's kernel (the one the batched refill of weight_cache uses, so that a value does not
    depend on how it was refilled) or, on a backend without it, ATen."""
    run = getattr(_backend.get(), 'weight_prep_batch', None)
    if run is not None and w.is_contiguous():
        return run('wsq', [(w.detach(),)])[0]
    return w.detach().pow(2).sum([2, 3])


class _WeightSq(Function):
    """[OC, IC, k, k] -> sum over the taps of W^2, [OC, IC].  The value only changes with the weight, so it is computed once per weight
    version (weight_cache) instead of in every forward pass of every iteration phase; the backward is the closed form 2 g W."""

    @staticmethod
    def forward(ctx, w):
        from . import weight_cache
        ctx.save_for_backward(w)
        return weight_cache.derive(w, ('wsq',), lambda: _wsq_value(w), recipe=('wsq',)).detach()

    @staticmethod
    def backward(ctx, g):
        w, = ctx.saved_tensors
        return (2.0 * g)[:, :, None, None] * w


class _WeightSqAll(Function):
    """_WeightSq of every demodulated layer at once: the values come out of the cache of derived weight forms (refilled for the whole
    network by one grouped launch per optimiser step), the gradients 2 g W of all layers are ONE launch (gc_weight_sq_bwd_grouped_f32)
    instead of two ATen passes per layer.  With grad mode on inside backward (orders above two) the ATen formula runs."""

    _fast = {}      # ids of the weights -> (cache generation, versions, storage addresses, the cached tensors)

    @staticmethod
    def forward(ctx, *ws):
        from . import weight_cache
        ctx.save_for_backward(*ws)
        # 18 cache look-ups cost ~1 ms of host time per generator pass; while nothing was dropped from the cache and no weight changed its
        # version or storage, the tensors found last time are still the cache's own
        key = tuple(id(w._base if w._base is not None else w) for w in ws)
        stamp = (weight_cache.generation[0], tuple(w._version for w in ws), tuple(w.data_ptr() for w in ws))
        hit = _WeightSqAll._fast.get(key)
        if hit is not None and hit[0] == stamp and weight_cache.ENABLED:
            return tuple(t.detach() for t in hit[1])
        outs = [weight_cache.derive(w, ('wsq',), lambda w=w: _wsq_value(w), recipe=('wsq',)) for w in ws]
        stamp = (weight_cache.generation[0], stamp[1], stamp[2])          # a refill may have dropped stale entries on the way
        if all(weight_cache._derived.get(t.data_ptr()) is not None for t in outs):
            _WeightSqAll._fast[key] = (stamp, outs)
        return tuple(t.detach() for t in outs)

    @staticmethod
    def backward(ctx, *gs):
        ws = ctx.saved_tensors
        be = _backend.get()
        if torch.is_grad_enabled() or getattr(be, 'weight_sq_bwd', None) is None or not all(w.is_contiguous() for w in ws):
            return tuple((2.0 * g)[:, :, None, None] * w if ctx.needs_input_grad[i] else None for i, (g, w) in enumerate(zip(gs, ws)))
        idx = [i for i in range(len(ws)) if ctx.needs_input_grad[i]]
        outs = be.weight_sq_bwd([ws[i] for i in idx], [gs[i].contiguous() for i in idx]) if idx else []
        res = [None] * len(ws)
        for i, o in zip(idx, outs):
            res[i] = o
        return tuple(res)


def weight_sq_all(weights):
    """[sum over the taps of W^2, [OC, IC]] for every [1, OC, IC, k, k] (or [OC, IC, k, k]) weight of the list."""
    return list(_backend.call(_WeightSqAll, *[w.view(w.shape[-4:]) for w in weights]))


def demod_coefficients(weight, s, scale, eps=1e-8):
    """d[b,oc] = rsqrt(sum_{ic,k} (scale * W[oc,ic,k] * s[b,ic])^2 + eps); weight is [1,OC,IC,k,k], s is [B,IC]."""
    wsq = _backend.call(_WeightSq, weight.view(weight.shape[1:]))       # [OC, IC]  (view, not weight[0]: its backward is free)
    # scale^2 * (s^2 @ wsq^T) + eps as ONE GEMM call (alpha, bias epilogue) in every direction of differentiation (op/linear.py)
    from .linear import equal_linear
    return torch.rsqrt(equal_linear(s.pow(2), wsq, _eps_vector(s, wsq.shape[0], eps), scale * scale, 1.0))


_EPS_VECTORS = {}


def _eps_vector(like, n, eps):
    key = (like.device, like.dtype, n, eps)
    t = _EPS_VECTORS.get(key)
    if t is None:
        t = _EPS_VECTORS[key] = torch.full((n,), eps, device=like.device, dtype=like.dtype)
    return t


class _SafeDiv(Function):
    """num / den over [B, C] vectors, a zero ``den`` (measure zero: the gradient it scales is zero as well) counting as one.
    One launch (gc_rows_sum_div_f32 with one chunk) instead of compare + select + divide; closed under differentiation."""

    @staticmethod
    def forward(ctx, num, den):
        out = _backend.get().rows_sum_div(num.contiguous().unsqueeze(-1), den.contiguous())
        ctx.save_for_backward(den, out)
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None
        den, out = ctx.saved_tensors
        gq = _backend.call(_SafeDiv, g, den)
        return (gq if ctx.needs_input_grad[0] else None), (-(gq * out) if ctx.needs_input_grad[1] else None)


class _SumDiv(Function):
    """partial [B, C, J] -> sum_J partial / den [B, C] (zero-safe as _SafeDiv): the second stage of a plane reduction and the division
    by the modulation / demodulation factor in one launch."""

    @staticmethod
    def forward(ctx, partial, den):
        out = _backend.get().rows_sum_div(partial.contiguous(), den.contiguous())
        ctx.save_for_backward(den, out)
        ctx.chunks = partial.shape[-1]
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None
        den, out = ctx.saved_tensors
        gq = _backend.call(_SafeDiv, g, den)
        return (gq.unsqueeze(-1).expand(*gq.shape, ctx.chunks) if ctx.needs_input_grad[0] else None), (-(gq * out) if ctx.needs_input_grad[1] else None)


class _PlaneDot(Function):
    """[B,C,H,W] x [B,C,H,W] -> [B,C] on gc_plane_dot_f32, divided by ``den`` [B,C] when given (zero-safe, in the reduction's second stage)."""

    @staticmethod
    def forward(ctx, a, b, den=None):
        out = _backend.get().plane_dot(_dense_or_pitched(a), _dense_or_pitched(b), den)
        ctx.has_den = den is not None
        ctx.save_for_backward(a, b, den if den is not None else a.new_empty(0), out if den is not None else a.new_empty(0))
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None
        a, b, den, out = ctx.saved_tensors
        gq = _backend.call(_SafeDiv, g, den) if ctx.has_den else g
        g4 = gq[:, :, None, None]
        return (g4 * b if ctx.needs_input_grad[0] else None), (g4 * a if ctx.needs_input_grad[1] else None), \
            (-(gq * out) if ctx.has_den and ctx.needs_input_grad[2] else None)


class _ModConv(Function):
    """y = so * gconv(si * x, w_t) [+ bias] [+ residual]; si / so may be None.

    bias [N] and residual (shaped like y) ride in the convolution's epilogue (gc_conv_epilogue): ToRGB's `+ self.bias` and
    `+ self.upsample(skip)` (gan_model.py:430-433) cost no pass of their own.  fork=True additionally returns x itself as a second
    output for the OTHER consumer of x (the StyledConv output feeds both ToRGB and the next up-sampling layer): that consumer's
    gradient then arrives here as the second output-gradient and is added inside the input-gradient convolution (its residual
    epilogue) instead of by autograd's separate elementwise add over the largest tensors of G."""

    @staticmethod
    def forward(ctx, x, w_t, si, so, geom, bias=None, residual=None, fork=False, derived=False):
        ctx.derived = derived        # a node of a backward pass (an input-gradient convolution of some other node), not of the network's forward pass
        ep = None
        if bias is not None or residual is not None:
            ep = (bias, None, None, 1.0, 1.0, False, None if residual is None else residual.contiguous())
        y = _backend.get().conv2d(_dense_or_pitched(x), w_t.contiguous(), None if si is None else si.contiguous(),
                                  None if so is None else so.contiguous(), geom, epilogue=ep)
        ctx.geom, ctx.in_hw = geom, (x.shape[2], x.shape[3])
        ctx.has_si, ctx.has_so, ctx.has_bias, ctx.has_res = si is not None, so is not None, bias is not None, residual is not None
        ctx.set_materialize_grads(False)        # an output nobody differentiated arrives as None, not as a tensor of zeros
        empty = x.new_empty(0)
        # the out_scale gradient needs the convolution part of y only: keep what was added on top of it
        keep_res = residual if (residual is not None and so is not None) else empty
        keep_bias = bias if (bias is not None and so is not None) else empty
        ctx.save_for_backward(x, w_t, si if si is not None else empty, so if so is not None else empty, y, keep_res, keep_bias)
        return (y, x.view_as(x)) if fork else y

    @staticmethod
    def backward(ctx, gy, gfork=None):
        x, w_t, si, so, y, res, bias = ctx.saved_tensors
        si = si if ctx.has_si else None
        so = so if ctx.has_so else None
        g = ctx.geom
        need = ctx.needs_input_grad
        gx = gw = gsi = gso = gb = gres = None
        if gy is None:                      # only the forked copy was used downstream (or nothing at all)
            return (gfork if need[0] else None), None, None, None, None, None, None, None, None
        need_si = ctx.has_si and need[2]
        need_so = ctx.has_so and need[3]
        fused = need[1] and _backend.want_param_grads() and _samples_route(x, gy, g, need_si, need_so, ctx.derived)
        if need[0] or (need_si and not fused):
            gx = _backend.call(_ModConv, gy, _adjoint_weight(w_t), so, si, _adjoint_geom(g, *ctx.in_hw), None, gfork if need[0] else None, False, True)
        if fused:
            gw, gsi, gso = _weight_and_scale_grads(x, gy, w_t, si, so, g, need_si, need_so)
            need_si = need_so = False
        elif need[1] and _backend.want_param_grads():
            gw = _mod_weight_grad(x, gy, si, so, g)
        if need_si:
            # d/dsi sees the convolution only, not the forked gradient that was added in the epilogue
            conv_part = gx if (gfork is None or not need[0]) else gx - gfork
            gsi = _backend.call(_PlaneDot, x, conv_part, si)
        if need_so:
            conv_part = y
            if ctx.has_res:
                conv_part = conv_part - res
            if ctx.has_bias:
                conv_part = conv_part - bias.reshape(1, -1, 1, 1)
            gso = _backend.call(_PlaneDot, gy, conv_part, so)
        if ctx.has_bias and need[5] and _backend.want_param_grads():
            from .fused_act import _channel_sum
            gb = _channel_sum(gy)
        if ctx.has_res and need[6]:
            gres = gy
        return (gx if need[0] else None), gw, gsi, gso, None, gb, gres, None, None


class _ModConvAct(Function):
    """out = gain * lrelu(so * gconv(si * x, w_t) + noise_w * noise + bias): modulated convolution, noise injection and
    FusedLeakyReLU (StyledConv, gan_model.py:402-408) as ONE kernel; the pre-activation tensor is never written.

    Backward: the activation gradient pass (which reads gy and `out` anyway) also returns the bias / noise-strength sums
    and -- through x_pre = lrelu^-1(out / gain) - bias - noise_w * noise -- the plane sums of g_pre * x_pre, i.e. the
    out_scale gradient, so neither the pre-activation nor a _PlaneDot over it is needed.  The rest is _ModConv's backward.
    """

    @staticmethod
    def forward(ctx, x, w_t, si, so, bias, noise, noise_w, geom, slope, gain):
        out = _backend.get().conv2d(x.contiguous(), w_t.contiguous(), None if si is None else si.contiguous(),
                                    None if so is None else so.contiguous(), geom,
                                    epilogue=(bias, None if noise is None else noise.contiguous(), noise_w, slope, gain, True))
        ctx.geom, ctx.in_hw, ctx.cfg = geom, (x.shape[2], x.shape[3]), (slope, gain)
        ctx.has = (si is not None, so is not None, bias is not None, noise is not None)
        empty = x.new_empty(0)
        ctx.save_for_backward(x, w_t, *[t if t is not None else empty for t in (si, so, bias, noise, noise_w)], out)
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, gy):
        from .fused_act import _BiasActGrad, _BiasActGradReduce
        x, w_t, si, so, bias, noise, noise_w, out = ctx.saved_tensors
        has_si, has_so, has_bias, has_noise = ctx.has
        si, so, bias = (si if has_si else None), (so if has_so else None), (bias if has_bias else None)
        noise, noise_w = (noise, noise_w) if has_noise else (None, None)
        g, (slope, gain) = ctx.geom, ctx.cfg
        need = ctx.needs_input_grad
        gx = gw = gsi = gso = gb = gnw = None
        if gy is None or not any(need[:7]):
            return (None,) * 10
        want_so = has_so and need[3]
        if not _backend.want_param_grads():
            need = list(need)
            need[1] = need[4] = need[6] = False          # weight, bias, noise strength
        if (has_bias and need[4]) or (has_noise and need[6]) or want_so:
            g_pre, psum, pdot, pself = _backend.call(_BiasActGradReduce, gy, out, noise, slope, gain, bias, noise_w, want_so)
            if has_bias and need[4]:
                gb = psum.sum((0, 2))
            if has_noise and need[6]:
                gnw = pdot.sum().reshape(noise_w.shape)
            if want_so:
                gso = _backend.call(_SumDiv, pself, so)
        else:
            g_pre = _backend.call(_BiasActGrad, gy, out, slope, gain)
        need_si = has_si and need[2]
        fused = need[1] and _samples_route(x, g_pre, g, need_si, False)
        if need[0] or (need_si and not fused):
            gx = _backend.call(_ModConv, g_pre, _adjoint_weight(w_t), so, si, _adjoint_geom(g, *ctx.in_hw), None, None, False, True)
        if fused:
            gw, gsi, _ = _weight_and_scale_grads(x, g_pre, w_t, si, so, g, True, False)
        elif need[1]:
            gw = _mod_weight_grad(x, g_pre, si, so, g)
        if need_si and not fused:
            gsi = _backend.call(_PlaneDot, x, gx, si)
        return (gx if need[0] else None), gw, gsi, gso, gb, None, gnw, None, None, None


def _swapped_geom(g, x):
    """The weight gradient of a transposed convolution: correlate gy (as the "input", decimated by `up`) with x (as the "output gradient")."""
    return ConvGeom(g.kh, g.kw, 1, g.up, g.kh - 1 - g.pad_y, g.kw - 1 - g.pad_x, x.shape[2], x.shape[3])


def _mod_weight_grad(x, gy, si, so, g):
    if g.up == 1:
        return _backend.call(_ModWGrad, x, gy, si, so, g)
    return _adjoint_weight(_backend.call(_ModWGrad, gy, x, so, si, _swapped_geom(g, x)))


class _ModWGrad(Function):
    """gw[t,k,n] = sum_{b,o} si[b,k] x[b,k,o*down+t-pad] so[b,n] gy[b,n,o]."""

    @staticmethod
    def forward(ctx, x, gy, si, so, geom):
        ctx.geom, ctx.in_hw = geom, (x.shape[2], x.shape[3])
        ctx.has_si, ctx.has_so = si is not None, so is not None
        empty = x.new_empty(0)
        ctx.save_for_backward(x, gy, si if si is not None else empty, so if so is not None else empty)
        ctx.set_materialize_grads(False)
        return _backend.get().conv2d_wgrad(_dense_or_pitched(x), _dense_or_pitched(gy), None if si is None else si.contiguous(),
                                           None if so is None else so.contiguous(), geom)

    @staticmethod
    def backward(ctx, ggw):
        if ggw is None:
            return None, None, None, None, None
        x, gy, si, so = ctx.saved_tensors
        si = si if ctx.has_si else None
        so = so if ctx.has_so else None
        g = ctx.geom
        gx = ggy = gsi = gso = None
        need_si = ctx.has_si and ctx.needs_input_grad[2]
        need_so = ctx.has_so and ctx.needs_input_grad[3]
        if ctx.needs_input_grad[0] or need_si:
            gx = _backend.call(_ModConv, gy, _adjoint_weight(ggw), so, si, _adjoint_geom(g, *ctx.in_hw), None, None, False, True)
        if ctx.needs_input_grad[1] or need_so:
            ggy = _backend.call(_ModConv, x, ggw.contiguous(), si, so, g, None, None, False, True)
        if need_si:
            gsi = _backend.call(_PlaneDot, x, gx, si)
        if need_so:
            gso = _backend.call(_PlaneDot, gy, ggy, so)
        return (gx if ctx.needs_input_grad[0] else None), (ggy if ctx.needs_input_grad[1] else None), gsi, gso, None


def modulated_conv2d(x, weight, s, demodulate=True, upsample=False, blur_kernel=None, blur_pad=None, padding=None, apply_blur=True,
                     bias=None, residual=None, fork=False, demod=None):
    """x [B,IC,H,W]; weight [1,OC,IC,k,k] (the reference parameter layout); s [B,IC] = modulation(style).

    plain:    conv2d(padding = k // 2)                                    gan_model.py:325-329
    upsample: conv_transpose2d(stride 2, padding 0) -> Blur(blur_pad)     gan_model.py:295-307
    """
    _, oc, ic, k, _ = weight.shape
    scale = 1.0 / math.sqrt(ic * k * k)
    # demod: the coefficients computed by the caller ahead of time (Generator.forward's style path on its side stream)
    d = (demod if demod is not None else demod_coefficients(weight, s, scale)) if demodulate else None
    if upsample:
        if bias is not None or residual is not None or fork:
            raise NotImplementedError('modulated_conv2d: bias / residual / fork are built for the plain branch only')
        w_t = kernel_layout(weight.view(oc, ic, k, k), scale, flip=True)                   # correlation form, [k,k,IC,OC]
        oh, ow = (x.shape[2] - 1) * 2 + k, (x.shape[3] - 1) * 2 + k
        y = _backend.call(_ModConv, x, w_t, s, d, ConvGeom(k, k, 2, 1, k - 1, k - 1, oh, ow))
        # apply_blur=False (StyledConv): the caller fuses the Blur with what follows and reads the (possibly row-pitched) tensor in place
        return upfirdn2d(y, blur_kernel, pad=blur_pad) if apply_blur else y
    pad = k // 2 if padding is None else padding
    w_t = kernel_layout(weight.view(oc, ic, k, k), scale)
    oh, ow = x.shape[2] + 2 * pad - k + 1, x.shape[3] + 2 * pad - k + 1
    if bias is None and residual is None and not fork:
        return _backend.call(_ModConv, x, w_t, s, d, ConvGeom(k, k, 1, 1, pad, pad, oh, ow))
    # plain branch only: `+ bias` / `+ residual` in the convolution's epilogue, fork = (y, x for its other consumer)
    return _backend.call(_ModConv, x, w_t, s, d, ConvGeom(k, k, 1, 1, pad, pad, oh, ow), None if bias is None else bias.reshape(-1).contiguous(), residual, bool(fork))


def modulated_conv2d_act(x, weight, s, bias, noise, noise_weight, demodulate=True, padding=None, negative_slope=0.2, act_scale=2 ** 0.5, demod=None):
    """modulated_conv2d (plain branch) -> + noise_weight * noise -> FusedLeakyReLU(bias), in one kernel launch."""
    _, oc, ic, k, _ = weight.shape
    scale = 1.0 / math.sqrt(ic * k * k)
    d = (demod if demod is not None else demod_coefficients(weight, s, scale)) if demodulate else None
    pad = k // 2 if padding is None else padding
    w_t = kernel_layout(weight.view(oc, ic, k, k), scale)
    oh, ow = x.shape[2] + 2 * pad - k + 1, x.shape[3] + 2 * pad - k + 1
    if noise.shape[0] != x.shape[0] or noise.numel() != x.shape[0] * oh * ow:
        raise ValueError(f'noise shape {tuple(noise.shape)} does not match the output [{x.shape[0]}, {oc}, {oh}, {ow}]')
    return _backend.call(_ModConvAct, x, w_t, s, d, bias.reshape(-1).contiguous(), noise, noise_weight.reshape(-1).contiguous(),
                             ConvGeom(k, k, 1, 1, pad, pad, oh, ow), float(negative_slope), float(act_scale))
