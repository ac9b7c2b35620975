"""GEMM-only autograd for the small dense layers of the style path.

EqualLinear (gan_model.py:171-202) is ``lr_mul * bias + scale * (x @ W^T)``; the mapping network, the 26 style modulations and the 18
demodulation sums of G are 52 such products on [B, 512] operands per forward pass.  torch.addmm fuses alpha / beta into the GEMM call on
the way forward, but autograd's formulas for addmm / mm multiply by alpha in a separate elementwise launch after every product on the
way back (and again in the second-order passes of the path-length regulariser).  The Functions here keep the scale inside the GEMM call in
every direction and are closed under differentiation.
"""
import os

import torch
from torch import autograd
from . import _backend

_SMALL_GEMM = os.environ.get('GANCONTROL_SMALL_GEMM', '1') != '0'

_ZERO = {}


def _zero1(like):
    key = (like.device, like.dtype)
    t = _ZERO.get(key)
    if t is None:
        t = _ZERO[key] = torch.zeros(1, device=like.device, dtype=like.dtype)
    return t


def _addmm(bias, a, b, beta, alpha):
    """beta * bias + alpha * (a @ b): gc_small_gemm_f32 when the INNER extent is tiny (the weight gradients of the style path, rank-B
    updates: a GEMM library is all latency there), torch.addmm otherwise (and on the CPU)."""
    if a.is_cuda:
        from . import _backend
        be = _backend.get()
        if _SMALL_GEMM and getattr(be, 'small_gemm_ok', None) is not None and be.small_gemm_ok(a, b) and (bias is None or (bias.dim() == 1 and bias.is_contiguous() and bias.dtype == torch.float32)):
            return be.small_gemm(a, b, bias, beta, alpha)
    if bias is None:
        return torch.addmm(_zero1(a), a, b, beta=0, alpha=alpha)
    return torch.addmm(bias, a, b, beta=beta, alpha=alpha)


class _ScaledMM(autograd.Function):
    """alpha * (a @ b) as ONE GEMM call (the scale rides in the GEMM's alpha); closed under differentiation, so the backward and the
    second-order passes of the 34 EqualLinear layers of G are GEMM calls only -- autograd's own formulas for addmm / mm issue a separate
    elementwise multiply by alpha after every product."""

    @staticmethod
    def forward(ctx, a, b, alpha):
        ctx.save_for_backward(a, b)
        ctx.alpha = alpha
        return _addmm(None, a, b, 0.0, alpha)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = _backend.call(_ScaledMM, g, b.t(), ctx.alpha) if ctx.needs_input_grad[0] else None
        gb = _backend.call(_ScaledMM, a.t(), g, ctx.alpha) if ctx.needs_input_grad[1] else None
        return ga, gb, None


class _EqualLinearFn(autograd.Function):
    """beta * bias + alpha * (x @ W^T) in one GEMM call; backward = two _ScaledMM and a column sum."""

    @staticmethod
    def forward(ctx, x, weight, bias, alpha, beta):
        ctx.save_for_backward(x, weight)
        ctx.alpha, ctx.beta = alpha, beta
        return _addmm(bias, x, weight.t(), beta, alpha)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        gx = _backend.call(_ScaledMM, g, weight, ctx.alpha) if ctx.needs_input_grad[0] else None
        gw = _backend.call(_ScaledMM, g.t(), x, ctx.alpha) if ctx.needs_input_grad[1] else None
        gb = (g.sum(0) * ctx.beta if ctx.beta != 1 else g.sum(0)) if ctx.needs_input_grad[2] else None
        return gx, gw, gb, None, None




def scaled_mm(a, b, alpha):
    return _backend.call(_ScaledMM, a, b, float(alpha))


def equal_linear(x, weight, bias, alpha, beta):
    """beta * bias [N] + alpha * (x [M, K] @ weight[N, K]^T)."""
    return _backend.call(_EqualLinearFn, x, weight, bias, float(alpha), float(beta))
