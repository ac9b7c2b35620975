"""Test configuration: import paths, the ``gpu`` marker and the CPU emulation of the C ABI.

``-m "not gpu"`` runs here (no GPU): oracle vs golden vectors, host logic, the autograd wiring
over an emulated backend, ABI symbol checks and gloo world_size-2 runs.  ``-m gpu`` runs on the
MI355X box and calls the HIP kernels through the C ABI.  Nothing here reads /root/reference.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'gan-control_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, 'tests', 'golden')

# The oracle legs of the tests run on the host.  On a 256-thread GPU box PyTorch's default (one intra-op thread per hardware thread) makes
# the oracle's many small convolutions up to 60 x slower than 16 - 32 threads do (profiles/cpu_threads_r03.json): cap it.
torch.set_num_threads(max(1, min(torch.get_num_threads(), 32)))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def group(gold, prefix):
    """Entries 'prefix/key' of a fixture as torch tensors keyed by 'key'."""
    out = {}
    for k, v in gold.items():
        if k.startswith(prefix + '/'):
            out[k[len(prefix) + 1:]] = torch.from_numpy(v) if v.dtype.kind in 'fiu' else v
    return out


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


class EmulatedBackend:
    """CPU stand-in for libgancontrol_hip.so with the SAME primitive interface (tests only).

    Written from the formulas in include/gancontrol_hip.h with ATen ops, independently of both
    the oracle and the kernels, so the autograd layer can be validated without a GPU.
    """
    name = 'emulated'

    @staticmethod
    def _stuff_pad(x, up, pad_y0, pad_x0, ext_h, ext_w):
        n, c, h, w = x.shape
        u = x.new_zeros(n, c, h * up, w * up)
        u[:, :, ::up, ::up] = x
        u = F.pad(u, [pad_x0, 0, pad_y0, 0])
        u = F.pad(u, [0, ext_w - u.shape[3], 0, ext_h - u.shape[2]])
        return u

    def upfirdn2d(self, x, taps, up, down, pad_x0, pad_y0, out_h, out_w, flip):
        kh, kw = taps.shape
        u = self._stuff_pad(x, up, pad_y0, pad_x0, (out_h - 1) * down + kh, (out_w - 1) * down + kw)
        t = (torch.flip(taps, [0, 1]) if flip else taps).to(x.dtype)
        n, c = x.shape[:2]
        y = F.conv2d(u.reshape(n * c, 1, u.shape[2], u.shape[3]), t.reshape(1, 1, kh, kw), stride=down)
        return y.reshape(n, c, out_h, out_w)

    @staticmethod
    def upfirdn2d_act_supported(taps, up, down, out_h, out_w, planes):
        return tuple(taps.shape) == (4, 4) and up == 1 and down == 1 and out_w >= 64 and out_h >= 16 and planes <= 65535

    def upfirdn2d_act(self, x, taps, pad_x0, pad_y0, out_h, out_w, flip, bias, noise, noise_w, slope, gain):
        return self.bias_act(self.upfirdn2d(x, taps, 1, 1, pad_x0, pad_y0, out_h, out_w, flip), bias, noise, noise_w, slope, gain)

    # -- grouped dense layers (include/gancontrol_hip.h: gc_grouped_linear_*): flat layer-major tensors, see op/style.py
    @staticmethod
    def _xblock(x, batch, sp):
        return x[batch * sp.xcol: batch * (sp.xcol + sp.k)].view(batch, sp.k)

    def grouped_linear(self, x, batch, plan, weights, biases):
        out = []
        for sp, w, b in zip(plan.specs, weights, biases):
            y = sp.alpha * (self._xblock(x, batch, sp) @ w.t())
            if b is not None:
                y = y + sp.beta * b
            out.append(y.reshape(-1))
        return torch.cat(out) if out else x.new_zeros(0)

    def grouped_linear_bwd_x(self, gy, batch, plan, weights):
        gx = gy.new_zeros(batch * plan.in_cols)
        at = 0
        for sp, w in zip(plan.specs, weights):
            g = gy[batch * at: batch * (at + sp.n)].view(batch, sp.n)
            self._xblock(gx, batch, sp).copy_(sp.alpha * (g @ w))
            at += sp.n
        return gx

    def grouped_linear_bwd_w(self, gy, x, batch, plan, has_bias):
        gws, gbs, at = [], [], 0
        for sp, hb in zip(plan.specs, has_bias):
            g = gy[batch * at: batch * (at + sp.n)].view(batch, sp.n)
            gws.append(sp.alpha * (g.t() @ self._xblock(x, batch, sp)))
            gbs.append(sp.beta * g.sum(0) if hb else None)
            at += sp.n
        return gws, gbs

    @staticmethod
    def pw_act_supported(x, dy):
        return x.dim() == 4 and x.shape[1] <= 3

    def pw_act_wgrad(self, x, dy, y_ref, slope, gain):
        g = self.bias_act_bwd(dy, y_ref, slope, gain)
        dw = torch.einsum('bkp,bnp->kn', x.flatten(2), g.flatten(2))
        return dw.reshape(1, 1, *dw.shape), g.sum((0, 2, 3))

    def pw_act_dgrad(self, dy, y_ref, w_adj, slope, gain):
        g = self.bias_act_bwd(dy, y_ref, slope, gain)
        return torch.einsum('bnp,nk->bkp', g.flatten(2), w_adj[0, 0]).reshape(dy.shape[0], w_adj.shape[3], *dy.shape[2:])

    def upfirdn2d_actbwd(self, gy, y_ref, noise, taps, pad_x0, pad_y0, out_h, out_w, flip, slope, gain):
        g_pre = self.bias_act_bwd(gy, y_ref, slope, gain)
        b, c = gy.shape[:2]
        psum = g_pre.reshape(b, c, -1).sum(2, keepdim=True)
        pdot = (g_pre * noise.reshape(b, 1, *gy.shape[2:])).reshape(b, c, -1).sum(2, keepdim=True) if noise is not None else None
        return self.upfirdn2d(g_pre, taps, 1, 1, pad_x0, pad_y0, out_h, out_w, flip), psum, pdot

    def upfirdn2d_mask(self, x, taps, pad_x0, pad_y0, out_h, out_w, flip, mask_ref, slope, gain):
        return self.bias_act_bwd(self.upfirdn2d(x, taps, 1, 1, pad_x0, pad_y0, out_h, out_w, flip), mask_ref, slope, gain)

    def bias_act(self, x, bias, noise, noise_w, slope, gain):
        shape = [1, -1] + [1] * (x.ndim - 2)
        v = x
        if noise is not None:
            v = v + noise_w.reshape(-1)[0] * noise.reshape([x.shape[0], 1] + list(x.shape[2:]))
        if bias is not None:
            v = v + bias.reshape(shape)
        return F.leaky_relu(v, slope) * gain

    def bias_act_bwd(self, dy, y_ref, slope, gain):
        return dy * torch.where(y_ref > 0, torch.full_like(dy, gain), torch.full_like(dy, gain * slope))

    def bias_act_bwd_reduce(self, dy, y_ref, noise, slope, gain, self_dot=None):
        dx = self.bias_act_bwd(dy, y_ref, slope, gain)
        b, c = dy.shape[0], dy.shape[1]
        inner = dy.numel() // (b * c)
        chunk = 16384
        chunks = -(-inner // chunk)

        def chunked(t):
            return F.pad(t.reshape(t.shape[0], t.shape[1], inner), [0, chunks * chunk - inner]).reshape(t.shape[0], t.shape[1], chunks, chunk)

        flat = chunked(dx)
        psum = flat.sum(3)
        pdot = pself = None
        if noise is not None:
            pdot = (flat * chunked(noise.reshape(b, 1, inner))).sum(3)
        if self_dot is not None:
            bias, noise_w = self_dot
            pre = torch.where(y_ref > 0, y_ref / gain, y_ref / (gain * slope))
            if bias is not None:
                pre = pre - bias.reshape([1, -1] + [1] * (y_ref.ndim - 2))
            if noise is not None:
                pre = pre - noise_w * noise.reshape(b, 1, *y_ref.shape[2:])
            pself = (flat * chunked(pre)).sum(3)
        return dx, psum, pdot, pself

    def bias_act_bwd_reduce_adjoint(self, ggx, cs, cd, cw, y_ref, dx, noise, bias, noise_w, slope, gain, want_gyref):
        b, c = y_ref.shape[0], y_ref.shape[1]
        inner = y_ref.numel() // (b * c)
        chunk = 16384

        def spread(t):
            return t.repeat_interleave(chunk, dim=2)[:, :, :inner].reshape(y_ref.shape)

        nz = None if noise is None else noise.reshape(b, 1, *y_ref.shape[2:])
        total = torch.zeros_like(y_ref) if ggx is None else ggx.clone()
        if cs is not None:
            total = total + spread(cs)
        if cd is not None:
            total = total + spread(cd) * nz
        g_yref = pgb = pgn = None
        if cw is not None:
            w = spread(cw)
            pre = torch.where(y_ref > 0, y_ref / gain, y_ref / (gain * slope))
            if bias is not None:
                pre = pre - bias.reshape([1, -1] + [1] * (y_ref.ndim - 2))
            if nz is not None:
                pre = pre - noise_w * nz
            total = total + w * pre
            wg = w * dx
            if want_gyref:
                g_yref = torch.where(y_ref > 0, wg / gain, wg / (gain * slope))

            def chunk_sums(t):
                flat = F.pad(t.reshape(b, c, inner), [0, -(-inner // chunk) * chunk - inner])
                return flat.reshape(b, c, -1, chunk).sum(3)

            pgb = chunk_sums(wg)
            if nz is not None:
                pgn = chunk_sums(wg * nz)
        g_dy = total * torch.where(y_ref > 0, torch.full_like(y_ref, gain), torch.full_like(y_ref, gain * slope))
        return g_dy, g_yref, pgb, pgn

    def conv2d_bn_relu(self, x, w, scale, shift, stride, pad_y, pad_x, relu=True, out=None, chan_off=0):
        y = F.conv2d(x, w, None, stride, (pad_y, pad_x))
        if scale is not None:
            y = y * scale.reshape(1, -1, 1, 1)
        if shift is not None:
            y = y + shift.reshape(1, -1, 1, 1)
        y = F.relu(y) if relu else y
        if out is None:
            return y
        out[:, chan_off:chan_off + y.shape[1]] = y
        return out

    def pool2d(self, x, k, stride, pad, mode, out=None, chan_off=0):
        y = F.max_pool2d(x, k, stride, pad) if mode == 'max' else F.avg_pool2d(x, k, stride, pad, count_include_pad=False)
        if out is None:
            return y
        out[:, chan_off:chan_off + y.shape[1]] = y
        return out

    def global_avgpool(self, x):
        return x.mean((2, 3), keepdim=True)

    def resize_bilinear(self, x, out_h, out_w, mul=1.0, add=0.0):
        return F.interpolate(x, size=(out_h, out_w), mode='bilinear', align_corners=False) * mul + add

    def rows_sum_div(self, partial, den=None):
        out = partial.sum(-1)
        return out if den is None else out / torch.where(den == 0, torch.ones_like(den), den)

    def plane_dot(self, a, b, den=None):
        return self.rows_sum_div((a * b).reshape(a.shape[0], a.shape[1], -1), den)

    def channel_sum(self, x):
        return x.sum([d for d in range(x.ndim) if d != 1])

    def weight_layout(self, src, taps, k, n, src_stride, dst_shape, dst_stride, flip, scale):
        view = torch.as_strided(src.contiguous().reshape(-1), (taps, k, n), tuple(src_stride))
        if flip:
            view = view.flip(0)
        dst = torch.empty(dst_shape, dtype=src.dtype)
        torch.as_strided(dst.reshape(-1), (taps, k, n), tuple(dst_stride)).copy_(view * scale)
        return dst

    def weight_prep_batch(self, kind, items):
        """The grouped re-layout (gc_weight_layout_grouped_f32): the same values as one weight_layout call per item."""
        if kind == 'wsq':
            return [w.pow(2).sum([2, 3]) for (w,) in items]
        if kind != 'layout':
            raise ValueError(kind)
        return [self.weight_layout(*it) for it in items]

    def weight_sq_bwd(self, weights, grads):
        return [(2.0 * g)[:, :, None, None] * w for w, g in zip(weights, grads)]

    def affine_warp(self, x, mat, in_h, in_w, out_h, out_w, adjoint):
        def fwd(img):
            ox = torch.arange(out_w, dtype=img.dtype).view(1, 1, out_w)
            oy = torch.arange(out_h, dtype=img.dtype).view(1, out_h, 1)
            m = mat.to(img.dtype)
            sx = m[:, 0].view(-1, 1, 1) * ox + m[:, 1].view(-1, 1, 1) * oy + m[:, 2].view(-1, 1, 1)
            sy = m[:, 3].view(-1, 1, 1) * ox + m[:, 4].view(-1, 1, 1) * oy + m[:, 5].view(-1, 1, 1)
            grid = torch.stack([(2 * sx + 1) / in_w - 1, (2 * sy + 1) / in_h - 1], dim=-1)       # pixel -> normalised, align_corners=False
            return F.grid_sample(img, grid, mode='bilinear', padding_mode='zeros', align_corners=False)
        if not adjoint:
            return fwd(x)
        probe = torch.zeros(x.shape[0], x.shape[1], in_h, in_w, dtype=x.dtype, requires_grad=True)
        with torch.enable_grad():
            g, = torch.autograd.grad(fwd(probe), probe, x.detach())
        return g

    def reflect_pad(self, x, pads, adjoint, in_hw):
        if not adjoint:
            return F.pad(x, tuple(pads), mode='reflect')
        probe = torch.zeros(x.shape[0], x.shape[1], in_hw[0], in_hw[1], dtype=x.dtype, requires_grad=True)
        with torch.enable_grad():
            g, = torch.autograd.grad(F.pad(probe, tuple(pads), mode='reflect'), probe, x.detach())
        return g

    def conv2d(self, x, w_t, in_scale, out_scale, geom, epilogue=None):
        if in_scale is not None:
            x = x * in_scale[:, :, None, None]
        u = self._stuff_pad(x, geom.up, geom.pad_y, geom.pad_x, (geom.out_h - 1) * geom.down + geom.kh,
                            (geom.out_w - 1) * geom.down + geom.kw)
        y = F.conv2d(u, w_t.permute(3, 2, 0, 1), stride=geom.down)
        assert y.shape[2:] == (geom.out_h, geom.out_w)
        if out_scale is not None:
            y = y * out_scale[:, :, None, None]
        if epilogue is not None:
            bias, noise, noise_w, slope, gain, activate = epilogue[:6]
            if noise is not None:
                y = y + noise_w * noise.reshape(y.shape[0], 1, *y.shape[2:])
            if bias is not None:
                y = y + bias.reshape(1, -1, 1, 1)
            if activate:
                y = F.leaky_relu(y, slope) * gain
            if len(epilogue) > 6 and epilogue[6] is not None:
                y = y + epilogue[6]
        return y

    def conv2d_wgrad(self, x, dy, in_scale, out_scale, geom):
        assert geom.up == 1
        w = torch.zeros(geom.kh, geom.kw, x.shape[1], dy.shape[1], dtype=x.dtype, requires_grad=True)
        with torch.enable_grad():
            y = self.conv2d(x.detach(), w, in_scale, out_scale, geom)
            g, = torch.autograd.grad(y, w, dy.detach())
        return g

    # per-sample weight gradients (gc_conv2d_wgrad_samples_* / gc_wgrad_samples_contract_f32): every shape has the form here
    def conv2d_wgrad_samples_bytes(self, x, dy, geom):
        return 4

    def conv2d_wgrad_samples(self, x, dy, in_scale, out_scale, geom):
        per = [self.conv2d_wgrad(x[b:b + 1], dy[b:b + 1], None if in_scale is None else in_scale[b:b + 1],
                                 None if out_scale is None else out_scale[b:b + 1], geom) for b in range(x.shape[0])]
        dws = torch.stack(per)
        return dws.sum(0), dws

    def wgrad_samples_contract(self, dws, w, scale_a, scale_c, want_a=True, want_c=True):
        b, a, c = dws.shape[0], dws.shape[-2], dws.shape[-1]
        prod = (dws * w.unsqueeze(0)).reshape(b, -1, a, c)
        safe = lambda den: torch.where(den == 0, torch.ones_like(den), den)
        g_a = prod.sum((1, 3)) / (safe(scale_a) if scale_a is not None else 1.0) if want_a else None
        g_c = prod.sum((1, 2)) / (safe(scale_c) if scale_c is not None else 1.0) if want_c else None
        return g_a, g_c


@pytest.fixture
def emu_backend():
    from gan_control_amd.models.op import _backend
    prev = _backend._install_for_tests(EmulatedBackend())
    yield
    _backend._install_for_tests(prev)


@pytest.fixture
def bf16x3_mode():
    """The split-bf16 convolution arithmetic (bench.py's default) for the duration of a -m gpu test."""
    from gan_control_amd.models.op import _backend
    hip = _backend.get()
    prev, hip.conv_mode = hip.conv_mode, 'bf16x3'
    yield
    hip.conv_mode = prev


@pytest.fixture
def bf16_mode():
    """Plain bf16 products (bench.py --precision bf16, BASELINE config[1]'s literal arithmetic)."""
    from gan_control_amd.models.op import _backend
    hip = _backend.get()
    prev, hip.conv_mode = hip.conv_mode, 'bf16'
    yield
    hip.conv_mode = prev


def have_gpu():
    return torch.cuda.is_available()
