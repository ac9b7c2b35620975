"""Raw (non-differentiable) primitives: one Python function per C-ABI entry point.

The autograd layer in this package is written against the ``Backend`` interface.  The only
implementation shipped is ``HipBackend`` (ctypes -> libgancontrol_hip.so).  Tests install an
emulation of the same interface to validate the autograd wiring on machines without a GPU;
the product never does.
"""
import contextlib
import ctypes
import math
import os

_PITCHED_OUTPUT = os.environ.get('GANCONTROL_PITCHED_OUTPUT', '1') != '0'      # dev knob: 0 = dense rows everywhere
from collections import namedtuple

import torch

from ... import _lib
from . import weight_cache

# ctx.needs_input_grad of a custom Function is fixed at forward time: inside `autograd.grad(outputs, inputs=[...])` a backward
# cannot see that only the listed inputs are wanted.  R1 (gradient w.r.t. the real images) and the path-length regulariser
# (gradient w.r.t. the latents) would therefore compute -- and throw away -- every weight / bias / noise-strength gradient of
# the network in their first (create_graph) backward.  The two call sites wrap that call in `activation_grads_only()` and
# the Functions of this package skip the parameter gradients while it is active.
_ACT_GRADS_ONLY = [False]


class activation_grads_only:
    def __enter__(self):
        self.prev = _ACT_GRADS_ONLY[0]
        _ACT_GRADS_ONLY[0] = True

    def __exit__(self, *exc):
        _ACT_GRADS_ONLY[0] = self.prev


def want_param_grads():
    return not _ACT_GRADS_ONLY[0]


# The reference's leaky-ReLU backward reads the activation's INPUT to pick the slope, so in a double-backward every parameter
# upstream of an activation is part of the graph and receives an exactly-zero gradient "through the mask".  The fused ops
# take the mask from the OUTPUT and cut that path (None instead of zeros; the trainer restores the zeros, _fill_missing_grads).
# The trainer's dry run (generator_trainer.py:301-327: which parameters have grad None under the regularisers?) needs the
# reference's graph connectivity: inside `strict_zero_grads()` the activation backward hands on zero tensors instead of None.
_STRICT_ZERO = [False]


class strict_zero_grads:
    def __enter__(self):
        self.prev = _STRICT_ZERO[0]
        _STRICT_ZERO[0] = True

    def __exit__(self, *exc):
        _STRICT_ZERO[0] = self.prev


def strict_zeros():
    return _STRICT_ZERO[0]


# Row-pitched OUTPUTS (the aligned-row layout of the odd-width intermediates) are an internal optimisation: a tensor
# handed to a caller outside this package must be an ordinary dense tensor (reference-style code calls .view() on it).  The public
# entry points (upfirdn2d, conv_transpose2d) therefore run with the layout switched off unless the call comes from a module of this
# package whose consumer reads the pitch (`_internal=True`); backward passes, whose tensors only travel between autograd nodes, keep it.
_PITCH_ALLOWED = [True]


class pitched_outputs:
    def __init__(self, allowed):
        self.allowed = bool(allowed)

    def __enter__(self):
        self.prev = _PITCH_ALLOWED[0]
        _PITCH_ALLOWED[0] = self.allowed

    def __exit__(self, *exc):
        _PITCH_ALLOWED[0] = self.prev


def pitch_allowed():
    return _PITCHED_OUTPUT and _PITCH_ALLOWED[0]


# Function.apply() costs ~15-25 us of host time per call (ctx, saved tensors, output wrapping) whether or not a graph is recorded, and one
# training iteration makes ~320 such calls going forward and ~500 more NESTED inside backward passes (every backward of this package is written
# with differentiable Functions so that R1 / path-length double-backward work).  When no graph CAN result -- grad mode is off (a plain backward),
# or no tensor argument requires a gradient (the generator forward of the D step, the discriminator of the G step's ... frozen parameters) --
# the Function's forward is called directly with a context that swallows the bookkeeping: same kernels, same values, none of the overhead.
class _NoGraphCtx:
    __slots__ = ('__dict__',)

    def save_for_backward(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass

    def mark_non_differentiable(self, *tensors):
        pass


_DIRECT = os.environ.get('GANCONTROL_DIRECT_FORWARD', '1') != '0'      # dev knob: 0 = always Function.apply


def call(fn, *args):
    """fn.apply(*args), or fn.forward(...) directly when autograd would record nothing."""
    if _DIRECT:
        if not torch.is_grad_enabled():
            return fn.forward(_NoGraphCtx(), *args)
        for a in args:
            if isinstance(a, torch.Tensor) and a.requires_grad:
                return fn.apply(*args)
        # Function.apply runs forward under no_grad: do the same, or an ATen op inside a forward that touches a tensor which is NOT an
        # argument (a root parameter behind the weight cache, module state held by a plan) would record a graph and hand back an output
        # with requires_grad set
        torch._C._set_grad_enabled(False)
        try:
            return fn.forward(_NoGraphCtx(), *args)
        finally:
            torch._C._set_grad_enabled(True)
    return fn.apply(*args)


# Geometry of the generalised convolution (gc_conv_desc minus batch/channels/in-size, which come from tensors)
ConvGeom = namedtuple('ConvGeom', 'kh kw up down pad_y pad_x out_h out_w')


class HipBackend:
    name = 'hip'
    timer = None          # optional utils.profiling.KernelTimer (bench.py); None in normal operation
    # 'f32': exact fp32 MFMA (the parity mode); 'bf16x3': split-bf16 MFMA, ~5e-6 relative error per layer;
    # 'bf16': one bf16 MFMA per product, fp32 accumulate and storage, ~3e-3 per layer (BASELINE config[1]; never the default)
    conv_mode = os.environ.get('GANCONTROL_CONV_PRECISION', 'f32')
    _conv_plans = {}      # (shape, geometry, mode) -> the library's per-shape answers (see conv2d)
    _chunks = {}          # plane size -> gc_bias_act_bwd_chunks

    def _bwd_chunks(self, inner):
        v = self._chunks.get(inner)
        if v is None:
            v = self._chunks[inner] = _lib.load().gc_bias_act_bwd_chunks(inner)
        return v

    @staticmethod
    def _guard(dev):
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        return torch.cuda.device(idx) if idx != torch.cuda.current_device() else None

    def _pitched_fir_ok(self, x, taps, up, down, out_h, out_w):
        return _lib.row_pitch(x) and self.upfirdn2d_act_supported(taps, up, down, out_h, out_w, x.shape[0] * x.shape[1])

    def upfirdn2d(self, x, taps, up, down, pad_x0, pad_y0, out_h, out_w, flip):
        """x [N,C,H,W] -> [N,C,out_h,out_w]; see gc_upfirdn2d_f32.  A row-pitched x (the output of a transposed convolution) is read in place
        by the 4x4 tile kernel and made contiguous for every other variant."""
        if _lib.row_pitch(x):
            if self._pitched_fir_ok(x, taps, up, down, out_h, out_w):
                return self.upfirdn2d_act(x, taps, pad_x0, pad_y0, out_h, out_w, flip, None, None, None, 1.0, 1.0, activate=False)
            x = x.contiguous()
        elif (pitch_allowed() and out_w % 4 != 0 and out_w >= 129 and self.conv_mode != 'f32'
              and self.upfirdn2d_act_supported(taps, up, down, out_h, out_w, x.shape[0] * x.shape[1])):
            # the (H + 1)-wide output of the Blur in front of a stride-2 convolution (and of the Blur adjoint in G): aligned rows for its
            # 16-byte stores (4.4 -> 5.6 TB/s); the stride-2 kernels read the pitch (gc_conv_desc.in_pitch)
            return self.upfirdn2d_act(x, taps, pad_x0, pad_y0, out_h, out_w, flip, None, None, None, 1.0, 1.0, activate=False,
                                      out_pitch=(out_w + 31) // 32 * 32)
        dev = _lib.require_cuda_f32(x, taps)
        n, c, h, w = x.shape
        y = torch.empty((n, c, out_h, out_w), dtype=x.dtype, device=dev)
        if y.numel() == 0:
            return y
        lib = _lib.load()
        g = self._guard(dev)
        t0 = None
        if self.timer:
            name = 'upfirdn2d_generic_kernel'          # mirrors the dispatch of gc_upfirdn2d_f32
            if tuple(taps.shape) == (4, 4) and n * c <= 65535:
                if up == 1 and down == 1 and out_w >= 64 and out_h >= 16:
                    name = 'fir44_tile_kernel'
                elif up == 1 and down == 1 and (out_h + 3) * (out_w + 3) <= 1296:
                    name = 'fir44_small_kernel'
                elif (up, down) == (1, 2) and out_w >= 32 and out_h >= 8:
                    name = 'fir44_down2_kernel'
                elif (up, down) == (2, 1) and out_w >= 32 and out_h >= 8:
                    name = 'fir44_up2_kernel'
            t0 = self.timer.start('fir44', name)
        if g: g.__enter__()
        try:
            rc = lib.gc_upfirdn2d_f32(_lib.ptr(x), _lib.ptr(taps), _lib.ptr(y), n * c, h, w, out_h, out_w,
                                      taps.shape[0], taps.shape[1], up, up, down, down, pad_x0, pad_y0, int(flip), _lib.stream_of(x))
        finally:
            if g: g.__exit__(None, None, None)
        _lib.check(rc, 'gc_upfirdn2d_f32')
        if t0 is not None:
            self.timer.stop(name, t0, 4.0 * (x.numel() + y.numel()))
        return y

    @staticmethod
    def upfirdn2d_act_supported(taps, up, down, out_h, out_w, planes):
        """Shapes gc_upfirdn2d_act_f32 takes (the 4x4 tile kernel); everything else runs FIR and activation as two launches."""
        return tuple(taps.shape) == (4, 4) and up == 1 and down == 1 and out_w >= 64 and out_h >= 16 and planes <= 65535

    def upfirdn2d_act(self, x, taps, pad_x0, pad_y0, out_h, out_w, flip, bias, noise, noise_w, slope, gain, activate=True, out_pitch=0):
        """gain * lrelu(FIR(x) + noise_w * noise + bias); see gc_upfirdn2d_act_f32 (gc_upfirdn2d_pitched_f32 for a row-pitched x)."""
        pitch = _lib.row_pitch(x)
        dev = _lib.require_cuda_f32(x, taps, bias, noise, noise_w, pitched=(x,))
        n, c, h, w = x.shape
        if out_pitch:
            y = torch.empty((n, c, out_h, out_pitch), dtype=x.dtype, device=dev)[..., :out_w]
        else:
            y = torch.empty((n, c, out_h, out_w), dtype=x.dtype, device=dev)
        if y.numel() == 0:
            return y
        lib = _lib.load()
        g = self._guard(dev)
        t0 = self.timer.start('fir44', 'fir44_tile_kernel') if self.timer else None
        if g: g.__enter__()
        try:
            if pitch or out_pitch or not activate:
                rc = lib.gc_upfirdn2d_pitched_f32(_lib.ptr(x), _lib.ptr(taps), _lib.ptr(y), n, c, h, w, pitch or w, out_h, out_w, out_pitch or out_w,
                                                  taps.shape[0], taps.shape[1],
                                                  pad_x0, pad_y0, int(flip), int(bool(activate)), _lib.ptr(bias), _lib.ptr(noise), _lib.ptr(noise_w),
                                                  float(slope), float(gain), _lib.stream_of(x))
            else:
                rc = lib.gc_upfirdn2d_act_f32(_lib.ptr(x), _lib.ptr(taps), _lib.ptr(y), n, c, h, w, out_h, out_w, taps.shape[0], taps.shape[1],
                                              pad_x0, pad_y0, int(flip), _lib.ptr(bias), _lib.ptr(noise), _lib.ptr(noise_w), float(slope), float(gain),
                                              _lib.stream_of(x))
        finally:
            if g: g.__exit__(None, None, None)
        _lib.check(rc, 'gc_upfirdn2d_act_f32')
        if t0 is not None:
            self.timer.stop('fir44_tile_kernel', t0, 4.0 * (x.numel() + y.numel()))
        return y

    def upfirdn2d_mask(self, x, taps, pad_x0, pad_y0, out_h, out_w, flip, mask_ref, slope, gain):
        """FIR(x) * (mask_ref > 0 ? gain : gain * slope) in one pass (gc_upfirdn2d_mask_f32: the Blur adjoint + the activation backward of the
        layer whose output the Blur read); shapes as upfirdn2d_act_supported, x may be row-pitched, mask_ref is [N, C, out_h, out_w] dense."""
        pitch = _lib.row_pitch(x)
        dev = _lib.require_cuda_f32(x, taps, mask_ref, pitched=(x,))
        n, c, h, w = x.shape
        if tuple(mask_ref.shape) != (n, c, out_h, out_w):
            raise RuntimeError('upfirdn2d_mask: mask reference %s, output [%d, %d, %d, %d]' % (tuple(mask_ref.shape), n, c, out_h, out_w))
        y = torch.empty((n, c, out_h, out_w), dtype=x.dtype, device=dev)
        if y.numel() == 0:
            return y
        t0 = self.timer.start('fir44', 'fir44_tile_kernel') if self.timer else None
        with (self._guard(dev) or contextlib.nullcontext()):
            rc = _lib.load().gc_upfirdn2d_mask_f32(_lib.ptr(x), _lib.ptr(taps), _lib.ptr(y), n, c, h, w, pitch or w, out_h, out_w, taps.shape[0], taps.shape[1],
                                                   pad_x0, pad_y0, int(flip), _lib.ptr(mask_ref), float(slope), float(gain), _lib.stream_of(x))
        _lib.check(rc, 'gc_upfirdn2d_mask_f32')
        if t0 is not None:
            self.timer.stop('fir44_tile_kernel', t0, 4.0 * (x.numel() + 2 * y.numel()))
        return y

    def upfirdn2d_actbwd(self, gy, y_ref, noise, taps, pad_x0, pad_y0, out_h, out_w, flip, slope, gain):
        """(gx, psum [B, C, tiles], pdot [B, C, tiles] | None): the Blur adjoint of gy * mask(y_ref) and the per-tile sums for the bias /
        noise-strength gradients in one pass over gy (gc_upfirdn2d_actbwd_f32).  gx comes back row-pitched when its width is odd."""
        dev = _lib.require_cuda_f32(gy, y_ref, noise, taps)
        n, c, h, w = gy.shape
        lib = _lib.load()
        tiles = lib.gc_upfirdn2d_actbwd_tiles(out_h, out_w)
        out_pitch = (out_w + 31) // 32 * 32 if (pitch_allowed() and out_w % 4 != 0 and out_w >= 129 and self.conv_mode != 'f32') else 0
        if out_pitch:
            gx = torch.empty((n, c, out_h, out_pitch), dtype=gy.dtype, device=dev)[..., :out_w]
        else:
            gx = torch.empty((n, c, out_h, out_w), dtype=gy.dtype, device=dev)
        psum = torch.empty((n, c, tiles), dtype=gy.dtype, device=dev)
        pdot = torch.empty((n, c, tiles), dtype=gy.dtype, device=dev) if noise is not None else None
        t0 = self.timer.start('fir44', 'fir44_tile_kernel') if self.timer else None
        with (self._guard(dev) or contextlib.nullcontext()):
            rc = lib.gc_upfirdn2d_actbwd_f32(_lib.ptr(gy), _lib.ptr(y_ref), _lib.ptr(noise), _lib.ptr(taps), _lib.ptr(gx), _lib.ptr(psum), _lib.ptr(pdot),
                                             n, c, h, w, out_h, out_w, out_pitch, taps.shape[0], taps.shape[1], pad_x0, pad_y0, int(flip),
                                             float(slope), float(gain), _lib.stream_of(gy))
        _lib.check(rc, 'gc_upfirdn2d_actbwd_f32')
        if t0 is not None:
            self.timer.stop('fir44_tile_kernel', t0, 4.0 * (2 * gy.numel() + gx.numel()))
        return gx, psum, pdot

    # -- backward of a fused (<= 3 -> n) 1x1 convolution + bias + leaky-ReLU (D's FromRGB layer) without an activation-backward pass -------
    @staticmethod
    def pw_act_supported(x, dy):
        """Shapes gc_pw_act_wgrad_f32 / gc_pw_act_dgrad_f32 take, where they pay: a bandwidth-sized plane and at most 3 input channels."""
        return x.dim() == 4 and x.shape[1] <= 3 and x.shape[0] <= 65535 and dy.shape[2] * dy.shape[3] * dy.shape[0] >= (1 << 18)

    def pw_act_wgrad(self, x, dy, y_ref, slope, gain):
        """(dw [1, 1, k, n], dbias [n]) of y = lrelu(conv1x1(x, w) + bias) * gain for the gradient dy at y (mask applied while dy is read)."""
        dev = _lib.require_cuda_f32(x, dy, y_ref)
        b, k, h, w = x.shape
        n = dy.shape[1]
        lib = _lib.load()
        nbytes = lib.gc_pw_act_wgrad_workspace(b, k, n, h * w)
        ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=dev)
        out = torch.empty((k + 1, n), dtype=x.dtype, device=dev)
        t0 = self.timer.start('wgrad') if self.timer else None
        with (self._guard(dev) or contextlib.nullcontext()):
            rc = lib.gc_pw_act_wgrad_f32(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(y_ref), _lib.ptr(out), b, k, n, h * w, float(slope), float(gain),
                                         _lib.ptr(ws), ws.numel() * 4, _lib.stream_of(x))
        _lib.check(rc, 'gc_pw_act_wgrad_f32')
        if t0 is not None:
            self.timer.stop('wgrad_mfma_kernel(+reduce)', t0, 2.0 * b * k * n * h * w)
        return out[:k].view(1, 1, k, n), out[k]

    def pw_act_dgrad(self, dy, y_ref, w_adj, slope, gain):
        """gx [B, k, H, W] = conv1x1(dy * mask(y_ref), w_adj) with w_adj [1, 1, n, k] (the input-gradient weights)."""
        dev = _lib.require_cuda_f32(dy, y_ref, w_adj)
        b, n, h, w = dy.shape
        k = w_adj.shape[3]
        gx = torch.empty((b, k, h, w), dtype=dy.dtype, device=dev)
        with (self._guard(dev) or contextlib.nullcontext()):
            rc = _lib.load().gc_pw_act_dgrad_f32(_lib.ptr(dy), _lib.ptr(y_ref), _lib.ptr(w_adj), _lib.ptr(gx), b, n, k, h * w, float(slope), float(gain),
                                                 _lib.stream_of(dy))
        _lib.check(rc, 'gc_pw_act_dgrad_f32')
        return gx

    def bias_act(self, x, bias, noise, noise_w, slope, gain):
        """y = gain * lrelu(x + bias[c] + noise_w * noise[b, :]); x is [B, C, *]."""
        dev = _lib.require_cuda_f32(x, bias, noise, noise_w)
        y = torch.empty_like(x)
        if x.numel() == 0:
            return y
        batch, ch = x.shape[0], x.shape[1]
        inner = x.numel() // (batch * ch)
        lib = _lib.load()
        g = self._guard(dev)
        t0 = self.timer.start('bias_act') if self.timer else None
        if g: g.__enter__()
        try:
            rc = lib.gc_bias_act_f32(_lib.ptr(x), _lib.ptr(bias), _lib.ptr(noise), _lib.ptr(noise_w), _lib.ptr(y),
                                     batch, ch, inner, slope, gain, _lib.stream_of(x))
        finally:
            if g: g.__exit__(None, None, None)
        _lib.check(rc, 'gc_bias_act_f32')
        if t0 is not None:
            self.timer.stop('bias_act_kernel', t0, 4.0 * (2 * x.numel() + ch + (noise.numel() if noise is not None else 0)))
        return y

    def bias_act_bwd(self, dy, y_ref, slope, gain):
        dev = _lib.require_cuda_f32(dy, y_ref)
        dx = torch.empty_like(dy)
        if dy.numel() == 0:
            return dx
        lib = _lib.load()
        g = self._guard(dev)
        if g: g.__enter__()
        try:
            rc = lib.gc_bias_act_bwd_f32(_lib.ptr(dy), _lib.ptr(y_ref), _lib.ptr(dx), dy.numel(), slope, gain, _lib.stream_of(dy))
        finally:
            if g: g.__exit__(None, None, None)
        _lib.check(rc, 'gc_bias_act_bwd_f32')
        return dx

    def bias_act_bwd_reduce(self, dy, y_ref, noise, slope, gain, self_dot=None):
        """-> (dx, psum [B, C, chunks], pdot [B, C, chunks] or None, pself [B, C, chunks] or None).

        self_dot = (bias [C] | None, noise_w [1] | None) additionally requests pself = chunk sums of dx * (pre-activation
        rebuilt from y_ref); see gc_bias_act_bwd_reduce_self_f32.
        """
        dev = _lib.require_cuda_f32(dy, y_ref, noise)
        batch, ch = dy.shape[0], dy.shape[1]
        inner = dy.numel() // (batch * ch)
        lib = _lib.load()
        chunks = self._bwd_chunks(inner)
        dx = torch.empty_like(dy)
        psum = torch.empty((batch, ch, chunks), dtype=dy.dtype, device=dev)
        pdot = torch.empty((batch, ch, chunks), dtype=dy.dtype, device=dev) if noise is not None else None
        pself = bias = noise_w = None
        if self_dot is not None:
            bias, noise_w = self_dot
            _lib.require_cuda_f32(dy, bias, noise_w)
            pself = torch.empty((batch, ch, chunks), dtype=dy.dtype, device=dev)
        g = self._guard(dev)
        if g: g.__enter__()
        try:
            rc = lib.gc_bias_act_bwd_reduce_self_f32(_lib.ptr(dy), _lib.ptr(y_ref), _lib.ptr(noise), _lib.ptr(bias), _lib.ptr(noise_w),
                                                     _lib.ptr(dx), _lib.ptr(psum), _lib.ptr(pdot), _lib.ptr(pself),
                                                     batch, ch, inner, slope, gain, _lib.stream_of(dy))
        finally:
            if g: g.__exit__(None, None, None)
        _lib.check(rc, 'gc_bias_act_bwd_reduce_self_f32')
        return dx, psum, pdot, pself

    def bias_act_bwd_reduce_adjoint(self, ggx, cs, cd, cw, y_ref, dx, noise, bias, noise_w, slope, gain, want_gyref):
        """Second-order pass through bias_act_bwd_reduce -> (g_dy, g_yref | None, pgb | None, pgn | None); see gc_bias_act_bwd_reduce_adjoint_f32."""
        dev = _lib.require_cuda_f32(y_ref, ggx, cs, cd, cw, dx, noise, bias, noise_w)
        batch, ch = y_ref.shape[0], y_ref.shape[1]
        inner = y_ref.numel() // (batch * ch)
        lib = _lib.load()
        chunks = self._bwd_chunks(inner)
        for t in (cs, cd, cw):
            if t is not None and tuple(t.shape) != (batch, ch, chunks):
                raise RuntimeError('bias_act_bwd_reduce_adjoint: cotangent shape %s, expected %s' % (tuple(t.shape), (batch, ch, chunks)))
        g_dy = torch.empty_like(y_ref)
        g_yref = torch.empty_like(y_ref) if (want_gyref and cw is not None) else None
        pgb = torch.empty((batch, ch, chunks), dtype=y_ref.dtype, device=dev) if cw is not None else None
        pgn = torch.empty((batch, ch, chunks), dtype=y_ref.dtype, device=dev) if (cw is not None and noise is not None) else None
        g = self._guard(dev)
        if g: g.__enter__()
        try:
            rc = lib.gc_bias_act_bwd_reduce_adjoint_f32(_lib.ptr(ggx), _lib.ptr(cs), _lib.ptr(cd), _lib.ptr(cw), _lib.ptr(y_ref), _lib.ptr(dx), _lib.ptr(noise),
                                                        _lib.ptr(bias), _lib.ptr(noise_w), _lib.ptr(g_dy), _lib.ptr(g_yref), _lib.ptr(pgb), _lib.ptr(pgn),
                                                        batch, ch, inner, slope, gain, _lib.stream_of(y_ref))
        finally:
            if g: g.__exit__(None, None, None)
        _lib.check(rc, 'gc_bias_act_bwd_reduce_adjoint_f32')
        return g_dy, g_yref, pgb, pgn

    # -- f-2: forward pass of the FID feature network (inference) ---------------------------------------
    def conv2d_bn_relu(self, x, w, scale, shift, stride, pad_y, pad_x, relu=True, out=None, chan_off=0):
        """relu?(scale * conv(x, w) + shift) with w in the reference layout [N, K, kh, kw]; writes channels [chan_off, chan_off + N) of
        ``out`` when given (a block's concatenated output), else returns a new tensor."""
        dev = _lib.require_cuda_f32(x, w, scale, shift, out)
        b, k, h, wd = x.shape
        n, _, kh, kw = w.shape
        oh, ow = (h + 2 * pad_y - kh) // stride + 1, (wd + 2 * pad_x - kw) // stride + 1
        if out is None:
            out = torch.empty((b, n, oh, ow), dtype=x.dtype, device=dev)
        if out.shape[0] != b or out.shape[2] != oh or out.shape[3] != ow or not out.is_contiguous():
            raise ValueError(f'conv2d_bn_relu: output {tuple(out.shape)} does not match [{b}, *, {oh}, {ow}]')
        rc = _lib.load().gc_conv2d_bn_relu_f32(_lib.ptr(x.contiguous()), _lib.ptr(w.contiguous()), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(out),
                                               b, k, n, h, wd, kh, kw, stride, pad_y, pad_x, int(bool(relu)), out.shape[1], chan_off, _lib.stream_of(x))
        _lib.check(rc, 'gc_conv2d_bn_relu_f32')
        return out

    def pool2d(self, x, k, stride, pad, mode, out=None, chan_off=0):
        """mode 'max' | 'avg' (average over the taps inside the image); same channel-offset output convention."""
        dev = _lib.require_cuda_f32(x, out)
        b, c, h, wd = x.shape
        oh, ow = (h + 2 * pad - k) // stride + 1, (wd + 2 * pad - k) // stride + 1
        if out is None:
            out = torch.empty((b, c, oh, ow), dtype=x.dtype, device=dev)
        rc = _lib.load().gc_pool2d_f32(_lib.ptr(x.contiguous()), _lib.ptr(out), b, c, h, wd, k, stride, pad, 0 if mode == 'max' else 1, out.shape[1], chan_off,
                                       _lib.stream_of(x))
        _lib.check(rc, 'gc_pool2d_f32')
        return out

    def global_avgpool(self, x):
        dev = _lib.require_cuda_f32(x)
        b, c = x.shape[0], x.shape[1]
        out = torch.empty((b, c, 1, 1), dtype=x.dtype, device=dev)
        _lib.check(_lib.load().gc_global_avgpool_f32(_lib.ptr(x.contiguous()), _lib.ptr(out), b * c, x.numel() // max(b * c, 1), _lib.stream_of(x)), 'gc_global_avgpool_f32')
        return out

    def resize_bilinear(self, x, out_h, out_w, mul=1.0, add=0.0):
        dev = _lib.require_cuda_f32(x)
        b, c, h, wd = x.shape
        out = torch.empty((b, c, out_h, out_w), dtype=x.dtype, device=dev)
        _lib.check(_lib.load().gc_resize_bilinear_f32(_lib.ptr(x.contiguous()), _lib.ptr(out), b * c, h, wd, out_h, out_w, float(mul), float(add), _lib.stream_of(x)),
                   'gc_resize_bilinear_f32')
        return out

    def small_gemm_ok(self, a, b):
        """True when alpha * (a @ b) + beta * bias is taken by gc_small_gemm_f32 (inner extent <= 8, output <= 2^19 elements)."""
        if not (a.is_cuda and b.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == 2 and b.dim() == 2):
            return False
        return bool(_lib.load().gc_small_gemm_ok(a.shape[0], a.shape[1], b.shape[1], b.stride(0), b.stride(1)))

    def small_gemm(self, a, b, bias, beta, alpha):
        """alpha * (a [M, K] @ b [K, N]) + beta * bias [N]; a and b may be transposed views (strides are passed on)."""
        dev = a.device
        if bias is not None:
            _lib.require_cuda_f32(bias)
        m, k, n = a.shape[0], a.shape[1], b.shape[1]
        out = torch.empty((m, n), dtype=torch.float32, device=dev)
        g = self._guard(dev)
        if g: g.__enter__()
        try:
            rc = _lib.load().gc_small_gemm_f32(_lib.ptr(a), a.stride(0), a.stride(1), _lib.ptr(b), b.stride(0), b.stride(1), _lib.ptr(bias),
                                               float(beta), float(alpha), _lib.ptr(out), m, k, n, _lib.stream_of(a))
        finally:
            if g: g.__exit__(None, None, None)
        _lib.check(rc, 'gc_small_gemm_f32')
        return out

    # -- the style path: every dense layer of one kind in ONE launch (op/style.py) ----------------------------------------
    @staticmethod
    def _glin_table(plan, batch, tag):
        """The gc_glin_group table of a plan for one batch size: extents / scales / offsets are fixed, only pointers change per call."""
        key = (tag, batch)
        hit = plan.cache.get(key)
        if hit is None:
            n = len(plan.specs)
            table = (_lib.GlinGroup * n)()
            xoff, yoff = [], []
            at = 0
            for i, sp in enumerate(plan.specs):
                g = table[i]
                g.n, g.k, g.x_stride, g.alpha, g.beta = sp.n, sp.k, sp.k, sp.alpha, sp.beta
                xoff.append(4 * batch * sp.xcol)
                yoff.append(4 * batch * at)
                at += sp.n
            hit = plan.cache[key] = (table, xoff, yoff)
        return hit

    def grouped_linear(self, x, batch, plan, weights, biases):
        """y (flat, blocks [batch, n_g] in group order) = alpha_g * x_g @ w_g^T + beta_g * bias_g; see gc_grouped_linear_f32."""
        dev = _lib.require_cuda_f32(x, *weights, *[b for b in biases if b is not None])
        y = torch.empty(batch * plan.out_cols, dtype=x.dtype, device=dev)
        table, xoff, yoff = self._glin_table(plan, batch, 'fwd')
        xp, yp = x.data_ptr(), y.data_ptr()
        for i in range(len(plan.specs)):
            g = table[i]
            g.x, g.y, g.w, g.bias = xp + xoff[i], yp + yoff[i], weights[i].data_ptr(), (biases[i].data_ptr() if biases[i] is not None else None)
        with (self._guard(dev) or contextlib.nullcontext()):
            rc = _lib.load().gc_grouped_linear_f32(table, len(plan.specs), batch, _lib.stream_of(x))
        _lib.check(rc, 'gc_grouped_linear_f32')
        return y

    def grouped_linear_bwd_x(self, gy, batch, plan, weights):
        """gx (flat, the plan's input layout) = alpha_g * gy_g @ w_g; input blocks no group reads are zero."""
        dev = _lib.require_cuda_f32(gy, *weights)
        gx = (torch.empty if plan.covers_input else torch.zeros)(batch * plan.in_cols, dtype=gy.dtype, device=dev)
        table, xoff, yoff = self._glin_table(plan, batch, 'bwd_x')
        xp, yp = gx.data_ptr(), gy.data_ptr()
        for i in range(len(plan.specs)):
            g = table[i]
            g.x, g.y, g.w, g.bias = xp + xoff[i], yp + yoff[i], weights[i].data_ptr(), None
        with (self._guard(dev) or contextlib.nullcontext()):
            rc = _lib.load().gc_grouped_linear_bwd_x_f32(table, len(plan.specs), batch, _lib.stream_of(gy))
        _lib.check(rc, 'gc_grouped_linear_bwd_x_f32')
        return gx

    def grouped_linear_bwd_w(self, gy, x, batch, plan, has_bias):
        """([gw_g [n_g, k_g]], [gbias_g [n_g] | None]): views of two flat buffers written by one launch."""
        dev = _lib.require_cuda_f32(gy, x)
        specs = plan.specs
        gw_flat = torch.empty(sum(sp.n * sp.k for sp in specs), dtype=gy.dtype, device=dev)
        gb_flat = torch.empty(sum(sp.n for sp, hb in zip(specs, has_bias) if hb), dtype=gy.dtype, device=dev) if any(has_bias) else None
        table, xoff, yoff = self._glin_table(plan, batch, 'bwd_w')
        xp, yp = x.data_ptr(), gy.data_ptr()
        gws, gbs, wo, bo = [], [], 0, 0
        for i, sp in enumerate(specs):
            g = table[i]
            gw = gw_flat[wo:wo + sp.n * sp.k].view(sp.n, sp.k)
            wo += sp.n * sp.k
            gb = None
            if has_bias[i]:
                gb = gb_flat[bo:bo + sp.n]
                bo += sp.n
            g.x, g.y, g.w, g.bias = xp + xoff[i], yp + yoff[i], gw.data_ptr(), (gb.data_ptr() if gb is not None else None)
            gws.append(gw)
            gbs.append(gb)
        with (self._guard(dev) or contextlib.nullcontext()):
            rc = _lib.load().gc_grouped_linear_bwd_w_f32(table, len(specs), batch, _lib.stream_of(gy))
        _lib.check(rc, 'gc_grouped_linear_bwd_w_f32')
        return gws, gbs

    def rows_sum_div(self, partial, den=None):
        """[..., J] -> [...]: sum over the last dim, divided by ``den`` (same leading shape; a zero divisor counts as one)."""
        dev = _lib.require_cuda_f32(partial, den)
        lead = partial.shape[:-1]
        rows, chunks = int(math.prod(lead)), int(partial.shape[-1])
        out = torch.empty(lead, dtype=partial.dtype, device=dev)
        if rows == 0:
            return out
        g = self._guard(dev)
        if g: g.__enter__()
        try:
            rc = _lib.load().gc_rows_sum_div_f32(_lib.ptr(partial), _lib.ptr(den), _lib.ptr(out), rows, chunks, _lib.stream_of(partial))
        finally:
            if g: g.__exit__(None, None, None)
        _lib.check(rc, 'gc_rows_sum_div_f32')
        return out

    def plane_dot(self, a, b, den=None):
        """[B, C, *] x [B, C, *] -> [B, C]: sum over the trailing dims of a * b (divided by den [B, C], a zero divisor counting as one)."""
        pa, pb = _lib.row_pitch(a), _lib.row_pitch(b)
        if pa or pb:
            return self._plane_dot_pitched(a, b, den, pa, pb)
        dev = _lib.require_cuda_f32(a, b)
        batch, ch = a.shape[0], a.shape[1]
        inner = a.numel() // (batch * ch)
        lib = _lib.load()
        chunks = self._bwd_chunks(inner)
        partial = torch.empty((batch, ch, chunks), dtype=a.dtype, device=dev)
        g = self._guard(dev)
        if g: g.__enter__()
        try:
            rc = lib.gc_plane_dot_f32(_lib.ptr(a), _lib.ptr(b), _lib.ptr(partial), batch * ch, inner, _lib.stream_of(a))
        finally:
            if g: g.__exit__(None, None, None)
        _lib.check(rc, 'gc_plane_dot_f32')
        if den is None and chunks == 1:
            return partial.reshape(batch, ch)
        return self.rows_sum_div(partial, None if den is None else den.contiguous())

    def _plane_dot_pitched(self, a, b, den, pa, pb):
        """plane_dot with one or both operands row-pitched (gc_plane_dot_pitched_f32); the other must be contiguous."""
        if not pa:
            a = a.contiguous()
        if not pb:
            b = b.contiguous()
        dev = _lib.require_cuda_f32(a, b, pitched=(a, b))
        batch, ch, rows, width = a.shape
        lib = _lib.load()
        chunks = lib.gc_plane_dot_pitched_chunks(rows)
        partial = torch.empty((batch, ch, chunks), dtype=a.dtype, device=dev)
        rc = lib.gc_plane_dot_pitched_f32(_lib.ptr(a), _lib.ptr(b), _lib.ptr(partial), batch * ch, rows, width, pa or width, pb or width, _lib.stream_of(a))
        _lib.check(rc, 'gc_plane_dot_pitched_f32')
        return self.rows_sum_div(partial, None if den is None else den.contiguous())

    def weight_prep_batch(self, kind, items):
        """The grouped forms of weight_layout / the bf16 pack for weight_cache's batched refill: one launch per kind.
        items: 'layout' -> (src, taps, k, n, src_stride, dst_shape, dst_stride, flip, scale); 'pack' -> (w_t, conv-desc fields)."""
        if not items:
            return []
        lib = _lib.load()
        dev = items[0][0].device
        outs = []
        with (self._guard(dev) or contextlib.nullcontext()):
            if kind == 'layout':
                table = (_lib.WLayoutGroup * len(items))()
                for g, (src, taps, k, n, src_stride, dst_shape, dst_stride, flip, scale) in zip(table, items):
                    _lib.require_cuda_f32(src)
                    dst = torch.empty(dst_shape, dtype=src.dtype, device=dev)
                    g.src, g.dst, g.taps, g.k, g.n, g.flip_taps, g.scale = src.data_ptr(), dst.data_ptr(), taps, k, n, int(bool(flip)), float(scale)
                    for i in range(3):
                        g.src_stride[i], g.dst_stride[i] = src_stride[i], dst_stride[i]
                    outs.append(dst)
                _lib.check(lib.gc_weight_layout_grouped_f32(table, len(items), _lib.stream_of(items[0][0])), 'gc_weight_layout_grouped_f32')
            elif kind == 'pack':
                table = (_lib.WPackGroup * len(items))()
                for g, (w_t, fields) in zip(table, items):
                    _lib.require_cuda_f32(w_t)
                    desc = _lib.ConvDesc(*fields)
                    pbytes = lib.gc_conv2d_bf16x3_packed_bytes(desc)
                    if pbytes == 0:      # a stale recipe (the shape has no packed form under this build): per-tensor make() decides
                        raise _lib.UnsupportedError('weight_prep_batch: descriptor %r takes no packed weights' % (tuple(fields),))
                    buf = torch.empty(pbytes // 4, dtype=torch.float32, device=dev)
                    g.desc, g.w, g.packed, g.packed_bytes = desc, w_t.data_ptr(), buf.data_ptr(), pbytes
                    outs.append(buf)
                _lib.check(lib.gc_conv2d_pack_weights_bf16x3_grouped(table, len(items), _lib.stream_of(items[0][0])), 'gc_conv2d_pack_weights_bf16x3_grouped')
            elif kind == 'wsq':
                table = (_lib.WsqGroup * len(items))()
                for g, (w,) in zip(table, items):
                    _lib.require_cuda_f32(w)
                    out = torch.empty(w.shape[:2], dtype=w.dtype, device=dev)
                    g.w, g.g, g.out, g.rows, g.taps = w.data_ptr(), None, out.data_ptr(), w.shape[0] * w.shape[1], w.numel() // (w.shape[0] * w.shape[1])
                    outs.append(out)
                _lib.check(lib.gc_weight_sq_grouped_f32(table, len(items), _lib.stream_of(items[0][0])), 'gc_weight_sq_grouped_f32')
            else:
                # a kind this backend has no grouped entry for: the one answer weight_cache's refill takes as "do it per tensor" 
                raise _lib.UnsupportedError('weight_prep_batch: no grouped form for %r' % (kind,))
        return outs

    def weight_sq_bwd(self, weights, grads):
        """[2 * w * g[:, :, None, None]] for every (w [N, K, kh, kw], g [N, K]) pair: one launch (gc_weight_sq_bwd_grouped_f32)."""
        dev = _lib.require_cuda_f32(*weights, *grads)
        table = (_lib.WsqGroup * len(weights))()
        outs = []
        for t, w, g in zip(table, weights, grads):
            out = torch.empty_like(w)
            t.w, t.g, t.out, t.rows, t.taps = w.data_ptr(), g.data_ptr(), out.data_ptr(), w.shape[0] * w.shape[1], w.numel() // (w.shape[0] * w.shape[1])
            outs.append(out)
        with (self._guard(dev) or contextlib.nullcontext()):
            rc = _lib.load().gc_weight_sq_bwd_grouped_f32(table, len(weights), _lib.stream_of(weights[0]))
        _lib.check(rc, 'gc_weight_sq_bwd_grouped_f32')
        return outs

    def weight_layout(self, src, taps, k, n, src_stride, dst_shape, dst_stride, flip, scale):
        """dst[t',k,n] = scale * src[t,k,n] between two strided weight layouts; see gc_weight_layout_f32."""
        dev = _lib.require_cuda_f32(src)
        dst = torch.empty(dst_shape, dtype=src.dtype, device=dev)
        if dst.numel() != taps * k * n or src.numel() != taps * k * n:
            raise RuntimeError('weight_layout: %d x %d x %d elements expected, got src %d / dst %d' % (taps, k, n, src.numel(), dst.numel()))
        lib = _lib.load()
        i64x3 = ctypes.c_int64 * 3
        g = self._guard(dev)
        if g: g.__enter__()
        try:
            rc = lib.gc_weight_layout_f32(_lib.ptr(src), _lib.ptr(dst), taps, k, n, ctypes.byref(i64x3(*src_stride)), ctypes.byref(i64x3(*dst_stride)),
                                          int(bool(flip)), float(scale), _lib.stream_of(src))
        finally:
            if g: g.__exit__(None, None, None)
        _lib.check(rc, 'gc_weight_layout_f32')
        return dst

    def channel_sum(self, x):
        """[B, C, *] -> [C]: sum over every dim but 1."""
        dev = _lib.require_cuda_f32(x)
        batch, ch = x.shape[0], x.shape[1]
        out = torch.empty(ch, dtype=x.dtype, device=dev)
        if x.numel() == 0:
            return out.zero_()
        inner = x.numel() // (batch * ch)
        lib = _lib.load()
        nbytes = lib.gc_channel_sum_workspace(batch, ch, inner)
        ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=dev)
        g = self._guard(dev)
        if g: g.__enter__()
        try:
            rc = lib.gc_channel_sum_f32(_lib.ptr(x), _lib.ptr(out), batch, ch, inner, _lib.ptr(ws), ws.numel() * 4, _lib.stream_of(x))
        finally:
            if g: g.__exit__(None, None, None)
        _lib.check(rc, 'gc_channel_sum_f32')
        return out

    def affine_warp(self, x, mat, in_h, in_w, out_h, out_w, adjoint):
        """Bilinear affine resampling (adjoint=False: [B,C,in_h,in_w] -> [B,C,out_h,out_w]) or its transpose; see gc_affine_warp_bilinear_f32."""
        dev = _lib.require_cuda_f32(x, mat)
        b, c = x.shape[0], x.shape[1]
        if tuple(mat.shape) != (b, 6):
            raise RuntimeError('affine_warp: mat must be [batch, 6], got %s' % (tuple(mat.shape),))
        want = (out_h, out_w) if adjoint else (in_h, in_w)
        if tuple(x.shape[2:]) != want:
            raise RuntimeError('affine_warp: input plane %s, expected %s' % (tuple(x.shape[2:]), want))
        y = torch.empty((b, c) + ((in_h, in_w) if adjoint else (out_h, out_w)), dtype=x.dtype, device=dev)
        if y.numel() == 0:
            return y
        lib = _lib.load()
        g = self._guard(dev)
        if g: g.__enter__()
        try:
            rc = lib.gc_affine_warp_bilinear_f32(_lib.ptr(x), _lib.ptr(mat), _lib.ptr(y), b, c, in_h, in_w, out_h, out_w, int(bool(adjoint)), _lib.stream_of(x))
        finally:
            if g: g.__exit__(None, None, None)
        _lib.check(rc, 'gc_affine_warp_bilinear_f32')
        return y

    def reflect_pad(self, x, pads, adjoint, in_hw):
        """F.pad(mode='reflect') of [B,C,in_h,in_w] by (left, right, top, bottom), or its adjoint (input: the padded-size gradient)."""
        dev = _lib.require_cuda_f32(x)
        left, right, top, bottom = pads
        h, w = in_hw
        b, c = x.shape[0], x.shape[1]
        y = torch.empty((b, c, h, w) if adjoint else (b, c, h + top + bottom, w + left + right), dtype=x.dtype, device=dev)
        lib = _lib.load()
        g = self._guard(dev)
        if g: g.__enter__()
        try:
            rc = lib.gc_reflect_pad_f32(_lib.ptr(x), _lib.ptr(y), b * c, h, w, left, right, top, bottom, int(bool(adjoint)), _lib.stream_of(x))
        finally:
            if g: g.__exit__(None, None, None)
        _lib.check(rc, 'gc_reflect_pad_f32')
        return y

    @staticmethod
    def _desc(x, n_out, geom):
        b, k, h, w = x.shape
        return _lib.ConvDesc(b, k, n_out, h, w, geom.out_h, geom.out_w, geom.kh, geom.kw, geom.up, geom.down, geom.pad_y, geom.pad_x)

    def conv2d(self, x, w_t, in_scale, out_scale, geom, epilogue=None):
        """x [B,K,H,W], w_t [kh,kw,K,N] -> [B,N,out_h,out_w]; see gc_conv2d_fused_f32.

        epilogue = (bias [N] | None, noise [B,1,oh,ow] | None, noise_w [1] | None, slope, gain, activate[, residual [B,N,oh,ow] | None]) or None.
        """
        lib = _lib.load()
        n_out = w_t.shape[3]
        desc = self._desc(x, n_out, geom)
        mode_id = {'f32': 0, 'bf16x3': 1, 'bf16': 2}.get(self.conv_mode, 0)
        # the library's per-shape answers (packed-weight bytes, split-K bytes, whether / how the output is row-pitched) do not change
        # between calls: each is asked once per (shape, mode, pitches) -- up to four ctypes round trips less per convolution on the host
        pkey = (x.shape[0], x.shape[1], n_out, x.shape[2], x.shape[3], geom, mode_id)

        def ask(what, fn, *extra):
            k = (what, pkey, desc.in_pitch, desc.out_pitch)
            v = self._conv_plans.get(k)
            if v is None:
                v = self._conv_plans[k] = fn(desc, *extra)
            return v
        in_pitch = _lib.row_pitch(x)
        if in_pitch:          # the pitched output of a Blur: the split-bf16 stride-2 kernel reads it in place
            if geom.down == 2 and lib.gc_conv2d_in_pitch_ok(desc, {'f32': 0, 'bf16x3': 1, 'bf16': 2}.get(self.conv_mode, 0), 0):
                desc.in_pitch = in_pitch
            else:
                x = x.contiguous()
        dev = _lib.require_cuda_f32(x, w_t, in_scale, out_scale, pitched=(x,))
        # Rows of a (2H + 1)-wide transposed-convolution output are never 16-byte aligned and their partial-line stores bound that kernel
        # (64 -> 32 @512^2: 365 us, 240 us with aligned rows): where the library says so the output is written with a row pitch that is
        # a multiple of 32 floats and handed on as a strided view; its consumers (the Blur that follows, the plane reductions) read the
        # pitch, anything else makes it contiguous.
        pitch = ask('out_pitch', lib.gc_conv2d_out_pitch, mode_id) if (geom.up == 2 and x.shape[0] > 0 and pitch_allowed()) else 0
        if pitch:
            desc.out_pitch = pitch
            y = torch.empty((x.shape[0], n_out, geom.out_h, pitch), dtype=x.dtype, device=dev)[..., :geom.out_w]
        else:
            y = torch.empty((x.shape[0], n_out, geom.out_h, geom.out_w), dtype=x.dtype, device=dev)
        if y.numel() == 0:
            return y
        ep = None
        if epilogue is not None:
            bias, noise, noise_w, slope, gain, activate = epilogue[:6]
            residual = epilogue[6] if len(epilogue) > 6 else None
            _lib.require_cuda_f32(x, bias, noise, noise_w, residual, pitched=(x,))
            if residual is not None and tuple(residual.shape) != tuple(y.shape):
                raise RuntimeError('conv2d epilogue: residual shape %s, output shape %s' % (tuple(residual.shape), tuple(y.shape)))
            if bias is not None and bias.numel() != n_out:
                raise RuntimeError('conv2d epilogue: bias has %d elements, expected %d' % (bias.numel(), n_out))
            if noise is not None and noise.numel() != x.shape[0] * geom.out_h * geom.out_w:
                raise RuntimeError('conv2d epilogue: noise has %d elements, expected %d' % (noise.numel(), x.shape[0] * geom.out_h * geom.out_w))
            ep = ctypes.byref(_lib.ConvEpilogue(_lib.ptr(bias), _lib.ptr(noise), _lib.ptr(noise_w), float(slope), float(gain), int(bool(activate)), _lib.ptr(residual)))
        ws = packed = None
        if self.conv_mode in ('bf16x3', 'bf16'):
            pbytes = ask('packed', lib.gc_conv2d_bf16x3_packed_bytes)
            if pbytes:
                # hi / lo split of the weights: once per (weight, optimiser step) when w_t derives from a parameter (weight_cache.py)
                def pack():
                    buf = torch.empty(pbytes // 4, dtype=torch.float32, device=dev)
                    with (self._guard(dev) or contextlib.nullcontext()):
                        _lib.check(lib.gc_conv2d_pack_weights_bf16x3(desc, _lib.ptr(w_t), _lib.ptr(buf), pbytes, _lib.stream_of(w_t)), 'gc_conv2d_pack_weights_bf16x3')
                    return buf
                packed = weight_cache.derive(w_t, ('pack_bf16x3',), pack, recipe=lambda: ('pack', tuple(getattr(desc, f) for f, _ in _lib.ConvDesc._fields_)))
                sbytes = ask('splitk', lib.gc_conv2d_bf16x3_splitk_bytes)          # K slices of a small-plane launch
                ws = torch.empty(sbytes // 4, dtype=torch.float32, device=dev) if sbytes else None
            else:
                nbytes = lib.gc_conv2d_bf16x3_workspace(desc)
                ws = torch.empty(max(nbytes // 4, 4), dtype=torch.float32, device=dev)
        elif self.conv_mode == 'f32':
            nbytes = ask('f32ws', lib.gc_conv2d_f32_workspace)          # K slices of a small-plane launch (planes <= 16 px wide), else 0
            ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev) if nbytes else None
        else:
            raise RuntimeError('GANCONTROL_CONV_PRECISION must be f32, bf16x3 or bf16, got %r' % self.conv_mode)
        g = self._guard(dev)
        t0 = None
        if self.timer:
            from ...utils.profiling import conv_variant, conv_flops
            tname = conv_variant(geom, n_out, x.shape[0], x.shape[1], self.conv_mode, in_hw=(x.shape[2], x.shape[3]))
            t0 = self.timer.start('conv', tname)
        if g: g.__enter__()
        try:
            if self.conv_mode == 'f32':
                rc = lib.gc_conv2d_fused_f32_ws(desc, _lib.ptr(x), _lib.ptr(w_t), _lib.ptr(in_scale), _lib.ptr(out_scale), ep, _lib.ptr(y),
                                                _lib.ptr(ws), ws.numel() * 4 if ws is not None else 0, _lib.stream_of(x))
            elif packed is not None:
                fn = lib.gc_conv2d_fused_bf16_packed_f32 if self.conv_mode == 'bf16' else lib.gc_conv2d_fused_bf16x3_packed_f32
                rc = fn(desc, _lib.ptr(x), _lib.ptr(w_t), _lib.ptr(packed), packed.numel() * 4, _lib.ptr(in_scale),
                                                           _lib.ptr(out_scale), ep, _lib.ptr(y), _lib.ptr(ws), ws.numel() * 4 if ws is not None else 0, _lib.stream_of(x))
            else:
                rc = lib.gc_conv2d_fused_bf16x3_packed_f32(desc, _lib.ptr(x), _lib.ptr(w_t), None, 0, _lib.ptr(in_scale), _lib.ptr(out_scale), ep, _lib.ptr(y),
                                                           _lib.ptr(ws), ws.numel() * 4, _lib.stream_of(x))
        finally:
            if g: g.__exit__(None, None, None)
        _lib.check(rc, 'gc_conv2d_fused_f32')
        if t0 is not None:
            self.timer.stop(tname, t0, conv_flops(x.shape[0], x.shape[1], n_out, x.shape[2], x.shape[3], geom))
        return y

    def conv2d_wgrad(self, x, dy, in_scale, out_scale, geom):
        in_pitch = _lib.row_pitch(x)
        if in_pitch or _lib.row_pitch(dy):
            n_out_ = dy.shape[1]
            ok = in_pitch and geom.down == 2 and _lib.load().gc_conv2d_in_pitch_ok(self._desc(x, n_out_, geom), {'f32': 0, 'bf16x3': 1, 'bf16': 2}.get(self.conv_mode, 0), 1)
            if not ok:
                x, in_pitch = x.contiguous(), 0
            dy = dy.contiguous()
        return self._conv2d_wgrad(x, dy, in_scale, out_scale, geom, in_pitch)

    def _conv2d_wgrad(self, x, dy, in_scale, out_scale, geom, in_pitch=0):
        """x [B,K,H,W], dy [B,N,out_h,out_w] -> dw [kh,kw,K,N]; see gc_conv2d_wgrad_f32."""
        dev = _lib.require_cuda_f32(x, dy, in_scale, out_scale, pitched=(x,))
        n_out = dy.shape[1]
        dw = torch.empty((geom.kh, geom.kw, x.shape[1], n_out), dtype=x.dtype, device=dev)
        desc = self._desc(x, n_out, geom)
        desc.in_pitch = in_pitch
        lib = _lib.load()
        fast = self.conv_mode in ('bf16x3', 'bf16')
        wkey = ('wgrad_ws', x.shape[0], x.shape[1], n_out, x.shape[2], x.shape[3], geom, fast, in_pitch)
        nbytes = self._conv_plans.get(wkey)
        if nbytes is None:
            nbytes = self._conv_plans[wkey] = (lib.gc_conv2d_wgrad_bf16x3_workspace if fast else lib.gc_conv2d_wgrad_workspace)(desc)
        ws = torch.empty(max(nbytes // 4, 4), dtype=torch.float32, device=dev)
        g = self._guard(dev)
        t0 = self.timer.start('wgrad') if self.timer else None
        if g: g.__enter__()
        try:
            fn = (lib.gc_conv2d_wgrad_bf16_f32 if self.conv_mode == 'bf16' else lib.gc_conv2d_wgrad_bf16x3_f32) if fast else lib.gc_conv2d_wgrad_f32
            rc = fn(desc, _lib.ptr(x), _lib.ptr(dy), _lib.ptr(in_scale), _lib.ptr(out_scale), _lib.ptr(dw),
                    _lib.ptr(ws), ws.numel() * 4, _lib.stream_of(x))
        finally:
            if g: g.__exit__(None, None, None)
        _lib.check(rc, 'gc_conv2d_wgrad_f32')
        if t0 is not None:
            from ...utils.profiling import conv_flops
            self.timer.stop('wgrad_mfma_kernel(+reduce)', t0, conv_flops(x.shape[0], x.shape[1], n_out, x.shape[2], x.shape[3], geom))
        return dw

    _MODES = {'f32': 0, 'bf16x3': 1, 'bf16': 2}

    def conv2d_wgrad_samples_bytes(self, x, dy, geom):
        """Scratch bytes of conv2d_wgrad_samples for these operands in the current arithmetic, 0 = this shape has no per-sample form."""
        n_out = dy.shape[1]
        key = ('wgrad_samples_ws', x.shape[0], x.shape[1], n_out, x.shape[2], x.shape[3], geom, self.conv_mode)
        nbytes = self._conv_plans.get(key)
        if nbytes is None:
            nbytes = self._conv_plans[key] = _lib.load().gc_conv2d_wgrad_samples_workspace(self._desc(x, n_out, geom), self._MODES.get(self.conv_mode, 0))
        return nbytes

    def conv2d_wgrad_samples(self, x, dy, in_scale, out_scale, geom):
        """-> (dw [kh,kw,K,N], dw_samples [B,kh,kw,K,N]): gc_conv2d_wgrad_samples_*; the caller has checked conv2d_wgrad_samples_bytes() > 0."""
        in_pitch = _lib.row_pitch(x)
        if in_pitch and geom.down != 2:
            x, in_pitch = x.contiguous(), 0
        if _lib.row_pitch(dy):
            dy = dy.contiguous()
        dev = _lib.require_cuda_f32(x, dy, in_scale, out_scale, pitched=(x,))
        n_out = dy.shape[1]
        nbytes = self.conv2d_wgrad_samples_bytes(x, dy, geom)
        if nbytes <= 0:
            raise ValueError('conv2d_wgrad_samples: no per-sample form for this shape (conv2d_wgrad_samples_bytes() == 0)')
        dw = torch.empty((geom.kh, geom.kw, x.shape[1], n_out), dtype=x.dtype, device=dev)
        dws = torch.empty((x.shape[0], geom.kh, geom.kw, x.shape[1], n_out), dtype=x.dtype, device=dev)
        desc = self._desc(x, n_out, geom)
        desc.in_pitch = in_pitch
        lib = _lib.load()
        ws = torch.empty(max(nbytes // 4, 4), dtype=torch.float32, device=dev)
        fn = {'bf16x3': lib.gc_conv2d_wgrad_samples_bf16x3_f32, 'bf16': lib.gc_conv2d_wgrad_samples_bf16_f32}.get(self.conv_mode, lib.gc_conv2d_wgrad_samples_f32)
        g = self._guard(dev)
        t0 = self.timer.start('wgrad') if self.timer else None
        if g: g.__enter__()
        try:
            rc = fn(desc, _lib.ptr(x), _lib.ptr(dy), _lib.ptr(in_scale), _lib.ptr(out_scale), _lib.ptr(dw), _lib.ptr(dws),
                    _lib.ptr(ws), ws.numel() * 4, _lib.stream_of(x))
        finally:
            if g: g.__exit__(None, None, None)
        _lib.check(rc, 'gc_conv2d_wgrad_samples_f32')
        if t0 is not None:
            from ...utils.profiling import conv_flops
            self.timer.stop('wgrad_mfma_kernel(+reduce)', t0, conv_flops(x.shape[0], x.shape[1], n_out, x.shape[2], x.shape[3], geom))
        return dw, dws

    def wgrad_samples_contract(self, dws, w, scale_a, scale_c, want_a=True, want_c=True):
        """dws [B,T..,A,C], w [T..,A,C] -> (g_a [B,A] or None, g_c [B,C] or None): gc_wgrad_samples_contract_f32."""
        dev = _lib.require_cuda_f32(dws, w, scale_a, scale_c)
        b, a, c = dws.shape[0], dws.shape[-2], dws.shape[-1]
        taps = w.numel() // (a * c)
        if dws.numel() != b * w.numel() or not dws.is_contiguous() or not w.is_contiguous():
            raise ValueError(f'wgrad_samples_contract: shapes {tuple(dws.shape)} / {tuple(w.shape)}')
        g_a = torch.empty((b, a), dtype=dws.dtype, device=dev) if want_a else None
        g_c = torch.empty((b, c), dtype=dws.dtype, device=dev) if want_c else None
        lib = _lib.load()
        ws = torch.empty(max(b * a * c, 4), dtype=torch.float32, device=dev) if want_c else None
        g = self._guard(dev)
        if g: g.__enter__()
        try:
            rc = lib.gc_wgrad_samples_contract_f32(_lib.ptr(dws), _lib.ptr(w), _lib.ptr(scale_a), _lib.ptr(scale_c), _lib.ptr(g_a), _lib.ptr(g_c),
                                                   b, taps, a, c, _lib.ptr(ws), ws.numel() * 4 if ws is not None else 0, _lib.stream_of(dws))
        finally:
            if g: g.__exit__(None, None, None)
        _lib.check(rc, 'gc_wgrad_samples_contract_f32')
        return g_a, g_c


_active = HipBackend()
weight_cache.batch_runner = lambda: getattr(_active, 'weight_prep_batch', None)
weight_cache.pack_wanted = lambda: getattr(_active, 'conv_mode', 'f32') != 'f32'


def get():
    return _active


def _install_for_tests(backend):
    """Swap the primitive implementation (tests only: CPU emulation of the C ABI)."""
    global _active
    prev, _active = _active, backend
    return prev
