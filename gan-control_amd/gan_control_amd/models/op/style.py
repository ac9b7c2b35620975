"""The style path of the generator as a handful of launches.

Reference: ModulatedConv2d.forward gan_model.py:281-293 -- per layer ``style = self.modulation(style)`` (an EqualLinear, :171-202) and,
when demodulating, ``rsqrt((weight ** 2).sum([2, 3, 4]) + 1e-8)`` of the per-sample modulated weight.  A 1024 x 1024 generator has 26
such layers (18 of them demodulated): 26 + 3 * 18 launches of a few microseconds on [B, 512] operands per forward pass, three times that
in the backward pass, and again in the second-order pass of the path-length regulariser.

Here all layers travel together in LAYER-MAJOR FLAT tensors: a flat buffer is the concatenation over layers g of contiguous [B, C_g]
blocks.  Elementwise steps (square, rsqrt) are one ATen call over the flat buffer for all layers; the dense steps are
``grouped_linear`` -- one launch of gc_grouped_linear_f32 for all layers -- whose three primitives (forward, input gradient, weight
gradient) are each other's derivatives, so the Functions below close under differentiation (R1 never touches them; the path-length
regulariser differentiates them twice).  Per-layer [B, C_g] tensors are views of the flat buffers (``split``: one ``cat`` backward).
"""
import torch
from torch.autograd import Function

from . import _backend


class GroupSpec:
    """One dense layer of a grouped launch: y[b, :n] = alpha * x[b, :k] @ w[n, k]^T + beta * bias[n].

    ``xcol``: per-sample column offset of the layer's [B, k] block inside the flat input (blocks are contiguous: the block starts at element
    ``B * xcol``).  Output blocks are packed in group order."""
    __slots__ = ('n', 'k', 'alpha', 'beta', 'xcol')

    def __init__(self, n, k, alpha, beta, xcol):
        self.n, self.k, self.alpha, self.beta, self.xcol = int(n), int(k), float(alpha), float(beta), int(xcol)


class Plan:
    """A fixed list of GroupSpec plus the per-sample column counts of the flat input / output."""

    def __init__(self, specs, in_cols=None):
        self.specs = list(specs)
        self.out_cols = sum(s.n for s in self.specs)
        self.in_cols = int(in_cols) if in_cols is not None else max((s.xcol + s.k for s in self.specs), default=0)
        self.covers_input = sorted((s.xcol, s.xcol + s.k) for s in self.specs) == self._tiling()
        self.cache = {}          # scratch of the backend (ctypes tables per batch size)

    def _tiling(self):
        out, at = [], 0
        for s in sorted(self.specs, key=lambda s: s.xcol):
            out.append((at, at + s.k))
            at += s.k
        return out if at == self.in_cols else None

    def ycols(self):
        out, at = [], 0
        for s in self.specs:
            out.append(at)
            at += s.n
        return out


def _supported(plan):
    return all(s.k % 4 == 0 and s.xcol % 4 == 0 for s in plan.specs)


class _GLFwd(Function):
    """y = grouped_linear(x; w_0 .. w_{G-1}; bias_0 .. bias_{G-1}) over flat layer-major tensors (bias entries may be None)."""

    @staticmethod
    def forward(ctx, x, plan, batch, *wb):
        g = len(plan.specs)
        weights, biases = wb[:g], wb[g:]
        ctx.plan, ctx.batch = plan, batch
        ctx.has_bias = [b is not None for b in biases]
        ctx.save_for_backward(x, *weights)
        return _backend.get().grouped_linear(x.contiguous(), batch, plan, [w.contiguous() for w in weights], list(biases))

    @staticmethod
    def backward(ctx, gy):
        x, *weights = ctx.saved_tensors
        plan, batch = ctx.plan, ctx.batch
        g = len(plan.specs)
        need = ctx.needs_input_grad
        gx = _backend.call(_GLBwdX, gy, plan, batch, *weights) if need[0] else None
        gws, gbs = [None] * g, [None] * g
        want_w = any(need[3:3 + g]) or any(need[3 + g:])
        if want_w and _backend.want_param_grads():
            outs = _backend.call(_GLBwdW, gy, x, plan, batch, tuple(ctx.has_bias))
            gws = list(outs[:g])
            it = iter(outs[g:])
            gbs = [next(it) if hb else None for hb in ctx.has_bias]
        return (gx, None, None, *gws, *gbs)


class _GLBwdX(Function):
    """gx = alpha * gy @ w per group, scattered to the flat input layout (blocks no group reads stay zero)."""

    @staticmethod
    def forward(ctx, gy, plan, batch, *weights):
        ctx.plan, ctx.batch = plan, batch
        ctx.save_for_backward(gy, *weights)
        return _backend.get().grouped_linear_bwd_x(gy.contiguous(), batch, plan, [w.contiguous() for w in weights])

    @staticmethod
    def backward(ctx, ggx):
        gy, *weights = ctx.saved_tensors
        plan, batch = ctx.plan, ctx.batch
        g = len(plan.specs)
        need = ctx.needs_input_grad
        g_gy = _backend.call(_GLFwd, ggx, plan, batch, *weights, *([None] * g)) if need[0] else None
        gws = [None] * g
        if any(need[3:]) and _backend.want_param_grads():
            gws = list(_backend.call(_GLBwdW, gy, ggx, plan, batch, (False,) * g)[:g])
        return (g_gy, None, None, *gws)


class _GLBwdW(Function):
    """(gw_0 .. gw_{G-1}, gbias of the groups that have one): gw = alpha * gy^T @ x, gbias = beta * sum_b gy."""

    @staticmethod
    def forward(ctx, gy, x, plan, batch, has_bias):
        ctx.plan, ctx.batch, ctx.has_bias = plan, batch, has_bias
        ctx.save_for_backward(gy, x)
        gws, gbs = _backend.get().grouped_linear_bwd_w(gy.contiguous(), x.contiguous(), batch, plan, has_bias)
        return (*gws, *[b for b in gbs if b is not None])

    @staticmethod
    def backward(ctx, *gg):
        gy, x = ctx.saved_tensors
        plan, batch = ctx.plan, ctx.batch
        g = len(plan.specs)
        ggw = list(gg[:g])
        it = iter(gg[g:])
        ggb = [next(it) if hb else None for hb in ctx.has_bias]
        g_gy = _backend.call(_GLFwd, x, plan, batch, *ggw, *ggb) if ctx.needs_input_grad[0] else None
        g_x = _backend.call(_GLBwdX, gy, plan, batch, *ggw) if ctx.needs_input_grad[1] else None
        return g_gy, g_x, None, None, None


def grouped_linear(x_flat, batch, plan, weights, biases):
    """All dense layers of ``plan`` in one launch.  x_flat: flat layer-major input (blocks [batch, k_g] at element batch * xcol_g);
    weights[g]: [n_g, k_g]; biases[g]: [n_g] or None.  Returns the flat layer-major output (blocks [batch, n_g] in group order)."""
    if len(weights) != len(plan.specs) or len(biases) != len(plan.specs):
        raise ValueError('grouped_linear: %d groups, %d weights, %d biases' % (len(plan.specs), len(weights), len(biases)))
    if x_flat.numel() != batch * plan.in_cols:
        raise ValueError('grouped_linear: input has %d elements, the plan expects %d x %d' % (x_flat.numel(), batch, plan.in_cols))
    return _backend.call(_GLFwd, x_flat, plan, batch, *weights, *biases)


def blocks(flat, batch, cols):
    """Per-layer [batch, c] views of a flat layer-major tensor (one split: its backward is a single cat)."""
    return [t.view(batch, c) for t, c in zip(flat.split([batch * c for c in cols]), cols)]
