"""Parity of the HIP kernels (through the C ABI) with the golden vectors and with the emulated
reference formulas at shapes that cross every tile boundary.  Needs an MI355X: ``-m gpu``."""
import pytest
import torch

import op_checks as oc
from conftest import EmulatedBackend, rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(scope='module', autouse=True)
def _native_loaded():
    from gan_control_amd import _lib
    _lib.load()          # fail loudly if the HIP library is missing: there is no fallback to hide behind
    assert torch.cuda.is_available()


@pytest.mark.parametrize('case', oc.names(oc.UPF))
def test_upfirdn2d_golden(case):
    oc.check_upfirdn2d(case, DEV)


@pytest.mark.parametrize('case', oc.names(oc.BA))
def test_bias_act_golden(case):
    oc.check_bias_act(case, DEV)


def test_noise_bias_act():
    oc.check_noise_bias_act(DEV)


@pytest.mark.parametrize('case', oc.names(oc.CV, 'conv_'))
def test_equal_conv_golden(case):
    oc.check_equal_conv(case, DEV)


@pytest.mark.parametrize('case', oc.names(oc.CV, 'mod_'))
def test_modulated_conv_golden(case):
    oc.check_modulated_conv(case, DEV)


def test_conv_functional():
    oc.conv_functional_checks(DEV)


def _be():
    from gan_control_amd.models.op import _backend
    assert _backend.get().name == 'hip'
    return _backend.get(), EmulatedBackend()


@pytest.mark.parametrize('shape,pad,flip', [
    ((2, 3, 65, 130), (1, 1), True), ((1, 2, 257, 257), (1, 1), True), ((1, 4, 128, 128), (2, 2), True),
    ((1, 4, 128, 128), (1, 1), False), ((3, 1, 33, 100), (2, 1), False), ((1, 1, 513, 513), (1, 1), True),
    ((1, 2, 64, 64), (2, 2), True), ((2, 2, 17, 64), (1, 1), True), ((1, 1, 1025, 67), (1, 1), True)])
def test_upfirdn2d_fast_path_shapes(shape, pad, flip):
    hip, emu = _be()
    gen = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=gen)
    k = torch.randn(4, 4, generator=gen)
    oh, ow = shape[2] + pad[0] + pad[1] - 3, shape[3] + pad[0] + pad[1] - 3
    ref = emu.upfirdn2d(x.double(), k.double(), 1, 1, pad[0], pad[0], oh, ow, flip)
    out = hip.upfirdn2d(x.to(DEV), k.to(DEV), 1, 1, pad[0], pad[0], oh, ow, flip)
    assert rel_err(out, ref) < 2e-6
    # a non-16-byte-aligned output base takes the scalar-store variant
    buf = torch.empty(out.numel() + 1, device=DEV)
    xs = torch.empty(x.numel() + 1, device=DEV)[1:].view_as(x).copy_(x)
    out2 = hip.upfirdn2d(xs, k.to(DEV), 1, 1, pad[0], pad[0], oh, ow, flip)
    assert torch.equal(out2, out)


@pytest.mark.parametrize('up,down,ksz', [(2, 1, 4), (1, 2, 4), (2, 1, 12), (1, 2, 12), (3, 2, 5), (1, 1, 3), (1, 1, 1)])
def test_upfirdn2d_generic_shapes(up, down, ksz):
    hip, emu = _be()
    gen = torch.Generator().manual_seed(up * 100 + down * 10 + ksz)
    x = torch.randn(2, 3, 37, 41, generator=gen)
    k = torch.randn(ksz, ksz, generator=gen)
    for p0, p1 in [(0, 0), (ksz // 2, ksz // 2), (ksz - 1, 1), (-1, 2)]:
        oh, ow = (37 * up + p0 + p1 - ksz) // down + 1, (41 * up + p0 + p1 - ksz) // down + 1
        if oh < 1 or ow < 1:
            continue
        for flip in (True, False):
            ref = emu.upfirdn2d(x.double(), k.double(), up, down, p0, p0, oh, ow, flip)
            out = hip.upfirdn2d(x.to(DEV), k.to(DEV), up, down, p0, p0, oh, ow, flip)
            assert rel_err(out, ref) < 2e-6, (p0, p1, flip)


@pytest.mark.parametrize('shape', [(4, 512, 32, 32), (3, 7, 33, 33), (2, 5, 16, 16), (2, 3, 17, 9), (1, 40, 4, 4), (5, 3, 5, 5), (1, 1, 1, 1), (2, 3, 30, 33), (1, 2, 36, 36)])
def test_upfirdn2d_small_plane_kernel(shape):
    """4x4 taps, up = down = 1 on planes that fit LDS whole (fir44_small_kernel): the 4^2 .. 32^2 blur layers of G and D."""
    hip, emu = _be()
    gen = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=gen)
    k = torch.randn(4, 4, generator=gen)
    for p0, p1 in [(2, 2), (1, 1), (2, 1), (3, 0), (0, 3), (3, 3), (-1, 4)]:
        oh, ow = shape[2] + p0 + p1 - 3, shape[3] + p0 + p1 - 3
        if oh < 1 or ow < 1:
            continue
        for flip in (True, False):
            ref = emu.upfirdn2d(x.double(), k.double(), 1, 1, p0, p0, oh, ow, flip)
            out = hip.upfirdn2d(x.to(DEV), k.to(DEV), 1, 1, p0, p0, oh, ow, flip)
            assert tuple(out.shape) == tuple(ref.shape)
            assert rel_err(out, ref) < 2e-6, (p0, p1, flip)
    # linearity in the taps and agreement with the wide-plane tile kernel on a plane both can take is covered by embedding:
    # a 33-wide plane inside a 64-wide zero plane must give the same numbers where the supports coincide
    if shape[3] <= 33 and shape[2] >= 16:
        wide = torch.zeros(shape[0], shape[1], shape[2], 64)
        wide[..., :shape[3]] = x
        a = hip.upfirdn2d(x.to(DEV), k.to(DEV), 1, 1, 2, 2, shape[2] + 1, shape[3] + 1, True)
        b = hip.upfirdn2d(wide.to(DEV), k.to(DEV), 1, 1, 2, 2, shape[2] + 1, 65, True)
        assert rel_err(a[..., :shape[3] - 2], b[..., :shape[3] - 2]) < 1e-6


@pytest.mark.parametrize('up,down', [(1, 2), (2, 1)])
@pytest.mark.parametrize('shape', [(2, 3, 64, 130), (1, 2, 257, 255), (1, 1, 33, 513), (3, 1, 128, 128), (1, 2, 16, 140)])
def test_upfirdn2d_resampling_tile_kernels(up, down, shape):
    """4x4 taps with up = 2 or down = 2 on planes wide enough for the tile kernels (fir44_down2 / fir44_up2)."""
    hip, emu = _be()
    gen = torch.Generator().manual_seed(sum(shape) + up)
    x = torch.randn(*shape, generator=gen)
    k = torch.randn(4, 4, generator=gen)
    for p0, p1 in [(1, 1), (2, 1), (2, 2), (0, 3), (3, 0), (-1, 2), (-2, -1)]:
        oh, ow = (shape[2] * up + p0 + p1 - 4) // down + 1, (shape[3] * up + p0 + p1 - 4) // down + 1
        if oh < 1 or ow < 1:
            continue
        for flip in (True, False):
            ref = emu.upfirdn2d(x.double(), k.double(), up, down, p0, p0, oh, ow, flip)
            out = hip.upfirdn2d(x.to(DEV), k.to(DEV), up, down, p0, p0, oh, ow, flip)
            assert out.shape == ref.shape
            assert rel_err(out, ref) < 2e-6, (p0, p1, flip)
    # misaligned input base
    xs = torch.empty(x.numel() + 1, device=DEV)[1:].view_as(x).copy_(x)
    oh, ow = (shape[2] * up + 2 - 4) // down + 1, (shape[3] * up + 2 - 4) // down + 1
    a = hip.upfirdn2d(xs, k.to(DEV), up, down, 1, 1, oh, ow, True)
    b = hip.upfirdn2d(x.to(DEV), k.to(DEV), up, down, 1, 1, oh, ow, True)
    assert torch.equal(a, b)


@pytest.mark.parametrize('n,k,kh', [(512, 512, 3), (40, 70, 3), (3, 32, 1), (32, 3, 1), (33, 1, 3), (1, 65, 3), (130, 64, 5)])
def test_weight_layout_kernel(n, k, kh):
    """gc_weight_layout_f32 against strided torch views: parameter -> kernel layout, adjoint, and back; bit-exact for scale 1."""
    hip, emu = _be()
    gen = torch.Generator().manual_seed(n * 7 + k)
    taps = kh * kh
    w = torch.randn(n, k, kh, kh, generator=gen)
    # (logical K, logical N, src strides for (tap, k, n), flip, scale)
    specs = [(k, n, (1, taps, k * taps), False, 1.0),        # [N,K,kh,kw] parameter -> [t,k,n]
             (k, n, (1, taps, k * taps), True, 0.37),        # ... mirrored, scaled
             (n, k, (1, k * taps, taps), True, 1.0)]         # the same buffer read as [K',N',kh,kw] (conv_transpose2d layout)
    for kk, nn, src_stride, flip, scale in specs:
        dst_shape, dst_stride = (kh, kh, kk, nn), (kk * nn, nn, 1)
        ref = emu.weight_layout(w, taps, kk, nn, src_stride, dst_shape, dst_stride, flip, scale)
        out = hip.weight_layout(w.to(DEV), taps, kk, nn, src_stride, dst_shape, dst_stride, flip, scale)
        assert torch.equal(out.cpu(), ref), (src_stride, flip, scale)
    w_t = torch.randn(kh, kh, k, n, generator=gen)
    ref = w_t.flip(0, 1).transpose(2, 3).contiguous()
    out = hip.weight_layout(w_t.to(DEV), taps, k, n, (k * n, n, 1), (kh, kh, n, k), (n * k, 1, k), True, 1.0)
    assert torch.equal(out.cpu(), ref)
    back = hip.weight_layout(out, taps, n, k, (n * k, k, 1), (kh, kh, k, n), (k * n, 1, n), True, 1.0)
    assert torch.equal(back.cpu(), w_t)


def test_weight_layout_autograd_closure():
    from gan_control_amd.models.op.weight_layout import kernel_layout, adjoint_layout
    w = torch.randn(24, 40, 3, 3, device=DEV, dtype=torch.float32, requires_grad=True)
    out = kernel_layout(w, 0.5, flip=True)
    ref = (w * 0.5).flip(2, 3).permute(2, 3, 1, 0)
    assert torch.equal(out, ref.contiguous())
    g = torch.randn_like(out)
    gw, = torch.autograd.grad(out, w, g, create_graph=True)
    gref, = torch.autograd.grad(ref, w, g)
    assert torch.allclose(gw, gref, rtol=0, atol=0)
    adj = adjoint_layout(out)
    assert torch.equal(adj, out.flip(0, 1).transpose(2, 3).contiguous())
    gg, = torch.autograd.grad(adj, w, torch.ones_like(adj))
    assert torch.allclose(gg, torch.full_like(gg, 0.5))


@pytest.mark.parametrize('shape', [(2, 6, 40, 70), (1, 5, 129, 131), (3, 8, 16, 16)])
def test_bias_act_bwd_reduce_self_dot(shape):
    """gc_bias_act_bwd_reduce_self_f32: the plane sums of dx * (pre-activation rebuilt from the activation output)."""
    hip, emu = _be()
    gen = torch.Generator().manual_seed(shape[2])
    x = torch.randn(*shape, generator=gen)
    bias, nz, nw = torch.randn(shape[1], generator=gen), torch.randn(shape[0], 1, *shape[2:], generator=gen), torch.randn(1, generator=gen)
    dy = torch.randn(*shape, generator=gen)
    for use_noise in (True, False):
        n_, w_ = (nz, nw) if use_noise else (None, None)
        y = emu.bias_act(x.double(), bias.double(), None if n_ is None else n_.double(), None if w_ is None else w_.double(), 0.2, 2 ** 0.5)
        ref = emu.bias_act_bwd_reduce(dy.double(), y, None if n_ is None else n_.double(), 0.2, 2 ** 0.5,
                                      self_dot=(bias.double(), None if w_ is None else w_.double()))
        # the rebuilt pre-activation is x itself: pself must equal sum(dx * x)
        assert rel_err(ref[3].sum(2), (ref[0] * x.double()).reshape(shape[0], shape[1], -1).sum(2)) < 1e-12
        out = hip.bias_act_bwd_reduce(dy.to(DEV), y.float().to(DEV), None if n_ is None else n_.to(DEV), 0.2, 2 ** 0.5,
                                      self_dot=(bias.to(DEV), None if w_ is None else w_.to(DEV)))
        for a, b in zip(out, ref):
            if b is not None:
                assert rel_err(a, b) < 2e-5, use_noise


@pytest.mark.parametrize('mode', ['f32', 'bf16x3'])
def test_modulated_conv2d_act_matches_two_pass(mode):
    """One-launch StyledConv (conv + noise + bias + leaky-ReLU) vs modulated_conv2d -> FusedLeakyReLU: values and gradients up to second order."""
    from gan_control_amd.models.op import modulated_conv2d, modulated_conv2d_act, fused_noise_bias_act
    hip, _ = _be()
    prev, hip.conv_mode = hip.conv_mode, mode
    try:
        gen = torch.Generator().manual_seed(3)
        x = torch.randn(2, 24, 33, 40, generator=gen).to(DEV).requires_grad_(True)
        w = torch.randn(1, 40, 24, 3, 3, generator=gen).to(DEV).requires_grad_(True)
        s = (torch.randn(2, 24, generator=gen) + 1.5).to(DEV).requires_grad_(True)
        b = torch.randn(40, generator=gen).to(DEV).requires_grad_(True)
        nw = torch.randn(1, generator=gen).to(DEV).requires_grad_(True)
        nz = torch.randn(2, 1, 33, 40, generator=gen).to(DEV)
        probe = torch.randn(2, 40, 33, 40, generator=gen).to(DEV)
        outs = []
        for fused in (True, False):
            if fused:
                y = modulated_conv2d_act(x, w, s, b, nz, nw)
            else:
                y = fused_noise_bias_act(modulated_conv2d(x, w, s), b, nz, nw)
            leaves = [x, w, s, b, nw]
            g1 = torch.autograd.grad((y * probe).sum(), leaves, retain_graph=True)
            gs, gxx = torch.autograd.grad((y * probe).sum() + y.square().sum(), [s, x], create_graph=True)     # path-length style: grad w.r.t. the style
            g2 = torch.autograd.grad(gs.square().sum() + gxx.square().mean(), leaves)
            outs.append([y.detach(), *g1, gs.detach(), *g2])
        tol = 2e-5 if mode == 'f32' else 3e-4
        for i, (a, c) in enumerate(zip(*outs)):
            assert rel_err(a, c) < tol, i
    finally:
        hip.conv_mode = prev


@pytest.mark.parametrize('shape,pad', [((2, 5, 67, 131), (1, 1)), ((1, 3, 129, 129), (2, 2)), ((2, 4, 64, 64), (2, 1))])
def test_upfirdn2d_fused_activation(shape, pad):
    """gc_upfirdn2d_act_f32 == gc_upfirdn2d_f32 followed by gc_bias_act_f32, bit for bit; autograd equals the two-pass composition."""
    from gan_control_amd.models.op import upfirdn2d, upfirdn2d_bias_act, fused_noise_bias_act
    hip, _ = _be()
    gen = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=gen).to(DEV)
    k = torch.randn(4, 4, generator=gen).to(DEV)
    oh, ow = shape[2] + pad[0] + pad[1] - 3, shape[3] + pad[0] + pad[1] - 3
    bias, nz, nw = torch.randn(shape[1], generator=gen).to(DEV), torch.randn(shape[0], 1, oh, ow, generator=gen).to(DEV), torch.randn(1, generator=gen).to(DEV)
    plain = hip.upfirdn2d(x, k, 1, 1, pad[0], pad[0], oh, ow, True)
    assert torch.equal(hip.upfirdn2d_act(x, k, pad[0], pad[0], oh, ow, True, bias, nz, nw, 0.2, 2 ** 0.5), hip.bias_act(plain, bias, nz, nw, 0.2, 2 ** 0.5))
    assert torch.equal(hip.upfirdn2d_act(x, k, pad[0], pad[0], oh, ow, True, bias, None, None, 0.2, 2 ** 0.5), hip.bias_act(plain, bias, None, None, 0.2, 2 ** 0.5))
    xs = [x.clone().requires_grad_(True) for _ in range(2)]
    bs = [bias.clone().requires_grad_(True) for _ in range(2)]
    ws = [nw.clone().requires_grad_(True) for _ in range(2)]
    res = []
    for i, fused in enumerate((True, False)):
        y = upfirdn2d_bias_act(xs[i], k, pad, bs[i], nz, ws[i]) if fused else fused_noise_bias_act(upfirdn2d(xs[i], k, pad=pad), bs[i], nz, ws[i])
        g1 = torch.autograd.grad(y.square().mean(), [xs[i], bs[i], ws[i]], retain_graph=True)
        gx, = torch.autograd.grad(y.sum() + y.square().sum(), xs[i], create_graph=True)
        g2 = torch.autograd.grad(gx.square().sum(), [xs[i], bs[i], ws[i]], allow_unused=True)
        res.append([y.detach(), *g1, gx.detach()] + [t for t in g2 if t is not None])
    assert len(res[0]) == len(res[1])
    for a, c in zip(*res):
        assert rel_err(a, c) < 1e-5


@pytest.mark.parametrize('mode', ['f32', 'bf16x3'])
def test_resblock_fused_sums_match_elementwise(mode):
    """ResBlock with both of its sums inside convolution epilogues (residual / fork) vs the plain elementwise version:
    values, first-order gradients and the R1-style second-order gradients."""
    from gan_control_amd.models import gan_model
    hip, _ = _be()
    prev, hip.conv_mode = hip.conv_mode, mode
    torch.manual_seed(5)
    blk = gan_model.ResBlock(24, 40).to(DEV)
    for prm in blk.parameters():
        prm.data.normal_()
    x0 = torch.randn(2, 24, 66, 70, device=DEV)
    res = []
    try:
        for fused in (True, False):
            gan_model._FUSE_EPILOGUE = fused
            x = x0.clone().requires_grad_(True)
            y = blk(x)
            params = list(blk.parameters())
            g1 = torch.autograd.grad(y.square().mean(), [x] + params, retain_graph=True)
            gx, = torch.autograd.grad(y.sum(), x, create_graph=True)
            g2 = torch.autograd.grad(gx.square().sum(), params, allow_unused=True)
            # a parameter that reaches the double-backward only through an activation mask has a ZERO gradient; the fused ops
            # report it as None (the trainer restores the zeros, _fill_missing_grads): both spellings are the same number
            res.append([y.detach(), *g1, gx.detach()] + [torch.zeros_like(prm) if t is None else t for t, prm in zip(g2, params)])
    finally:
        gan_model._FUSE_EPILOGUE = True
        hip.conv_mode = prev
    assert len(res[0]) == len(res[1])
    tol = 2e-5 if mode == 'f32' else 3e-4
    for i, (a, c) in enumerate(zip(*res)):
        if float(c.abs().max()) == 0.0:
            assert float(a.abs().max()) == 0.0, i
        else:
            assert rel_err(a, c) < tol, i


CONV_CASES = [
    # b, K, N, h, w, k, up, down, pad
    (2, 8, 8, 4, 4, 3, 1, 1, 1), (2, 16, 130, 8, 8, 3, 1, 1, 1), (1, 40, 64, 16, 16, 3, 1, 1, 1),
    (2, 32, 32, 40, 70, 3, 1, 1, 1), (1, 64, 64, 33, 65, 3, 1, 1, 1), (1, 130, 140, 20, 36, 3, 1, 1, 1),
    (2, 3, 32, 64, 64, 1, 1, 1, 0), (2, 32, 3, 50, 50, 1, 1, 1, 0), (1, 513, 40, 4, 4, 3, 1, 1, 1),
    (2, 24, 40, 33, 33, 3, 1, 2, 0), (1, 64, 130, 65, 129, 3, 1, 2, 0), (2, 16, 24, 31, 31, 1, 1, 2, 0),
    (2, 9, 70, 9, 9, 3, 1, 2, 0), (1, 8, 8, 17, 17, 3, 1, 2, 0),
    (2, 16, 24, 4, 4, 3, 2, 1, 2), (1, 40, 33, 8, 8, 3, 2, 1, 2), (2, 32, 32, 16, 16, 3, 2, 1, 2),
    (1, 64, 40, 32, 48, 3, 2, 1, 2), (1, 12, 70, 65, 40, 3, 2, 1, 2), (2, 6, 5, 16, 16, 1, 2, 1, 0),
    (1, 8, 8, 30, 30, 3, 2, 1, 0), (1, 8, 8, 30, 30, 3, 1, 1, 0), (1, 8, 8, 30, 30, 3, 1, 1, 2),
]


def _out_size(n, k, up, down, pad, transposed):
    if up > 1:
        return (n - 1) * up + k - 2 * (k - 1 - pad)
    return (n + 2 * pad - k) // down + 1


# output planes of <= 8 x 8 pixels (conv_f32_small_kernel, exact fp32 in every mode: K >= 64, N >= 64, batch * pixels <= 512): the networks' 4^2 / 8^2
# layers, odd planes, a pixel count that ends inside a 32-pixel MFMA column block, one sample, 1x1 taps, the largest batch
SMALL_CASES = [(4, 512, 512, 4, 4, 3, 1, 1, 1), (4, 512, 512, 8, 8, 3, 1, 1, 1), (3, 80, 64, 5, 7, 3, 1, 1, 1), (1, 64, 128, 3, 3, 3, 1, 1, 1),
               (2, 96, 64, 8, 6, 1, 1, 1, 0), (8, 64, 192, 8, 8, 3, 1, 1, 1), (5, 128, 64, 1, 1, 3, 1, 1, 1),
               (200, 64, 64, 1, 1, 3, 1, 1, 1),       # many samples of 1 x 1 planes: the halo fills the LDS (152 KB), one chunk per workgroup
               # stride 2 without padding (D's down-sampling convolutions after the Blur: 17 -> 8, 9 -> 4), odd planes, 1x1 taps
               (4, 512, 512, 17, 17, 3, 1, 2, 0), (4, 512, 512, 9, 9, 3, 1, 2, 0), (2, 64, 64, 7, 11, 3, 1, 2, 0), (3, 128, 64, 9, 9, 1, 1, 2, 0),
               # ragged channel counts: 513 -> 512 (D's last block after the minibatch-stddev channel) and its input gradient 512 -> 513
               (4, 513, 512, 4, 4, 3, 1, 1, 1), (4, 512, 513, 4, 4, 3, 1, 1, 1), (2, 72, 100, 8, 8, 3, 1, 1, 1), (2, 100, 70, 15, 15, 3, 1, 2, 0)]
# round 5: transposed (up = 2) 3x3 onto planes <= 10 x 10 on a zero-stuffed plane (G's 4^2 -> 9^2 layer, the input gradient of D's 9 -> 4 convolution),
# odd planes, another padding; and batches whose planes do not fit the LDS at once, in sample groups (D's 17 -> 8 at B = 8; 4 + 3 samples; 9^2 at B = 8)
SMALL_UP_CASES = [(4, 512, 512, 4, 4, 3, 2, 1, 2), (3, 80, 64, 3, 4, 3, 2, 1, 2), (2, 64, 96, 4, 4, 3, 2, 1, 1), (1, 64, 64, 1, 1, 3, 2, 1, 2)]
SMALL_GROUP_CASES = [(8, 512, 512, 17, 17, 3, 1, 2, 0), (7, 512, 64, 17, 17, 3, 1, 2, 0), (8, 512, 512, 4, 4, 3, 2, 1, 2), (21, 64, 64, 8, 8, 3, 1, 1, 1)]
# round 5: the weight gradients of these planes run on wgrad_f32_small_kernel (3x3 taps, planes <= 8 x 8, at most two LDS-sized sample groups): every
# 3x3 case above with up = 1 reaches it through the wgrad half of the tests; these add two sample groups at stride 1 (8 x 8 planes, B = 8), a ragged second
# group (17 -> 8, B = 3: groups of 2 + 1), pixel counts that are not multiples of 16 and of 2, and a batch past the limit (back on the pixel-tile kernels)
SMALL_WGRAD_CASES = [(8, 512, 64, 8, 8, 3, 1, 1, 1), (3, 128, 96, 17, 17, 3, 1, 2, 0), (3, 64, 80, 3, 5, 3, 1, 1, 1), (1, 70, 64, 3, 3, 3, 1, 1, 1), (1, 64, 64, 5, 5, 3, 1, 2, 0),
                     (8, 128, 64, 17, 17, 3, 1, 2, 0)]
SMALL_CASES = SMALL_CASES + SMALL_UP_CASES + SMALL_GROUP_CASES + SMALL_WGRAD_CASES

@pytest.mark.parametrize('case', CONV_CASES + SMALL_CASES)
def test_conv2d_kernel(case):
    from gan_control_amd.models.op._backend import ConvGeom
    hip, emu = _be()
    b, K, N, h, w, k, up, down, pad = case
    gen = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(b, K, h, w, generator=gen)
    wt = torch.randn(k, k, K, N, generator=gen)
    si = torch.randn(b, K, generator=gen)
    so = torch.rand(b, N, generator=gen) + 0.5
    oh, ow = _out_size(h, k, up, down, pad, up > 1), _out_size(w, k, up, down, pad, up > 1)
    geom = ConvGeom(k, k, up, down, pad, pad, oh, ow)
    if case in SMALL_CASES:
        from gan_control_amd.utils.profiling import conv_variant
        assert conv_variant(geom, N, b, K, 'f32', (h, w)).startswith('conv_f32_small_kernel'), 'this shape is meant to reach the small-plane kernel'
    for use_scales in (False, True):
        a = (si, so) if use_scales else (None, None)
        ref = emu.conv2d(x.double(), wt.double(), *[None if t is None else t.double() for t in a], geom)
        out = hip.conv2d(x.to(DEV), wt.to(DEV), *[None if t is None else t.to(DEV) for t in a], geom)
        assert rel_err(out, ref) < 5e-6, use_scales
    if up == 1:
        dy = torch.randn(b, N, oh, ow, generator=gen)
        for use_scales in (False, True):
            a = (si, so) if use_scales else (None, None)
            ref = emu.conv2d_wgrad(x.double(), dy.double(), *[None if t is None else t.double() for t in a], geom)
            out = hip.conv2d_wgrad(x.to(DEV), dy.to(DEV), *[None if t is None else t.to(DEV) for t in a], geom)
            assert rel_err(out, ref) < 5e-6, ('wgrad', use_scales)


def test_conv2d_large_wgrad_splits():
    """Many pixel tiles per split and several splits: the deterministic two-stage reduction."""
    from gan_control_amd.models.op._backend import ConvGeom
    hip, emu = _be()
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(3, 64, 96, 96, generator=gen)
    dy = torch.randn(3, 64, 96, 96, generator=gen)
    geom = ConvGeom(3, 3, 1, 1, 1, 1, 96, 96)
    ref = emu.conv2d_wgrad(x.double(), dy.double(), None, None, geom)
    out = hip.conv2d_wgrad(x.to(DEV), dy.to(DEV), None, None, geom)
    assert rel_err(out, ref) < 5e-6
    assert torch.equal(out, hip.conv2d_wgrad(x.to(DEV), dy.to(DEV), None, None, geom)), 'wgrad must be run-to-run deterministic'


@pytest.mark.parametrize('shape', [(2, 6, 5, 7), (3, 10), (2, 3, 64, 64), (1, 5, 33, 31), (4, 512, 4, 4), (2, 32, 128, 128)])
def test_bias_act_and_channel_sum_kernels(shape):
    hip, emu = _be()
    gen = torch.Generator().manual_seed(len(shape) + shape[1])
    x = torch.randn(*shape, generator=gen)
    b = torch.randn(shape[1], generator=gen)
    for noise in (False, True):
        if noise and len(shape) != 4:
            continue
        nz = torch.randn(shape[0], 1, *shape[2:], generator=gen) if noise else None
        nw = torch.randn(1, generator=gen) if noise else None
        ref = emu.bias_act(x.double(), b.double(), None if nz is None else nz.double(), None if nw is None else nw.double(), 0.2, 2 ** 0.5)
        out = hip.bias_act(x.to(DEV), b.to(DEV), None if nz is None else nz.to(DEV), None if nw is None else nw.to(DEV), 0.2, 2 ** 0.5)
        assert rel_err(out, ref) < 2e-6
    y = emu.bias_act(x, b, None, None, 0.2, 2 ** 0.5)
    dy = torch.randn(*shape, generator=gen)
    assert rel_err(hip.bias_act_bwd(dy.to(DEV), y.to(DEV), 0.2, 2 ** 0.5), emu.bias_act_bwd(dy, y, 0.2, 2 ** 0.5)) < 2e-6
    assert rel_err(hip.channel_sum(x.to(DEV)), emu.channel_sum(x.double())) < 1e-5


def test_reductions_beyond_65535_planes():
    """batch * channels > 65535 (a per-GPU mini-batch of 64 through D's 512-channel layers with fake and real interleaved): the
    activation-backward reductions, plane dots and channel sums run on 1-D grids with no plane limit."""
    hip, emu = _be()
    gen = torch.Generator().manual_seed(65536)
    shape = (137, 512, 2, 3)                      # 70144 planes
    y = torch.randn(*shape, generator=gen)
    dy = torch.randn(*shape, generator=gen)
    nz = torch.randn(shape[0], 1, 2, 3, generator=gen)
    bias, nw = torch.randn(512, generator=gen), torch.randn(1, generator=gen)
    ref = emu.bias_act_bwd_reduce(dy.double(), y.double(), nz.double(), 0.2, 2 ** 0.5, self_dot=(bias.double(), nw.double()))
    out = hip.bias_act_bwd_reduce(dy.to(DEV), y.to(DEV), nz.to(DEV), 0.2, 2 ** 0.5, self_dot=(bias.to(DEV), nw.to(DEV)))
    for a, c in zip(out, ref):
        assert rel_err(a, c) < 1e-5
    assert rel_err(hip.plane_dot(y.to(DEV), dy.to(DEV)), emu.plane_dot(y.double(), dy.double())) < 1e-5
    assert rel_err(hip.channel_sum(y.to(DEV)), emu.channel_sum(y.double())) < 1e-5
    cs = torch.randn(shape[0], 512, 1, generator=gen)
    ref = emu.bias_act_bwd_reduce_adjoint(dy.double(), cs.double(), None, None, y.double(), None, None, None, None, 0.2, 2 ** 0.5, False)
    out = hip.bias_act_bwd_reduce_adjoint(dy.to(DEV), cs.to(DEV), None, None, y.to(DEV), None, None, None, None, 0.2, 2 ** 0.5, False)
    assert rel_err(out[0], ref[0]) < 1e-5


@pytest.mark.parametrize('size', [32, 64, 256, 512, 1024])
def test_network_golden(size):
    """G and D forward against the reference at every BASELINE resolution (512: batch 2, 1024: batch 1), exact fp32 MFMA."""
    oc.check_network(size, DEV)


def test_error_reporting():
    from gan_control_amd.models.op._backend import ConvGeom
    hip, _ = _be()
    x = torch.zeros(1, 4, 8, 8, device=DEV)
    with pytest.raises(RuntimeError, match='taps'):
        hip.conv2d(x, torch.zeros(5, 5, 4, 4, device=DEV), None, None, ConvGeom(5, 5, 1, 1, 2, 2, 8, 8))
    with pytest.raises(RuntimeError):
        hip.upfirdn2d(torch.zeros(1, 1, 4, 4), torch.ones(2, 2), 1, 1, 0, 0, 3, 3, True)   # CPU tensor: refused


def test_step_golden():
    import step_checks
    step_checks.check_step(DEV)


BF16_CASES = CONV_CASES + [
    (2, 64, 64, 40, 70, 3, 1, 1, 1), (1, 130, 140, 33, 65, 3, 1, 1, 1), (1, 32, 32, 70, 40, 3, 1, 1, 1), (2, 48, 24, 36, 36, 1, 1, 1, 0),
    (1, 64, 130, 65, 129, 3, 1, 2, 0), (2, 40, 20, 71, 67, 3, 1, 2, 0), (2, 16, 24, 63, 63, 1, 1, 2, 0),
    (1, 64, 40, 32, 48, 3, 2, 1, 2), (2, 24, 70, 20, 33, 3, 2, 1, 2), (2, 16, 5, 40, 40, 1, 2, 1, 0), (1, 513, 40, 40, 40, 3, 1, 1, 1),
    (1, 32, 32, 30, 30, 3, 2, 1, 0), (1, 24, 40, 40, 33, 3, 2, 1, 1), (2, 64, 64, 64, 64, 3, 2, 1, 2), (1, 128, 32, 33, 70, 3, 2, 1, 2),
    (1, 64, 64, 63, 63, 1, 1, 2, 0), (2, 64, 96, 41, 77, 3, 1, 2, 0), (1, 128, 64, 129, 65, 3, 1, 2, 0), (1, 32, 32, 64, 96, 3, 1, 1, 1),
    (1, 32, 64, 65, 97, 3, 1, 2, 0), (2, 48, 70, 41, 77, 3, 1, 2, 0), (1, 40, 64, 63, 63, 1, 1, 2, 0), (1, 32, 64, 257, 259, 3, 1, 2, 0),
    (2, 3, 40, 67, 129, 1, 1, 1, 0), (2, 40, 3, 67, 129, 1, 1, 1, 0), (3, 64, 4, 128, 128, 1, 1, 1, 0), (1, 1, 70, 33, 35, 1, 1, 1, 0), (2, 4, 2, 20, 20, 1, 1, 1, 0),
    # planes large enough for the vector-ALU pointwise kernels (csrc/pointwise.hip): aligned and ragged
    (4, 32, 3, 256, 256, 1, 1, 1, 0), (4, 3, 32, 256, 256, 1, 1, 1, 0), (1, 3, 33, 515, 511, 1, 1, 1, 0), (2, 20, 2, 363, 365, 1, 1, 1, 0),
]


# shapes the wave-specialised kernel takes (conv_bf16x3_ws_kernel: K % 16 == 0, K >= 64, N % 64 == 0, >= 192 tiles x samples x oc blocks):
# several tiles per workgroup, odd chunk counts, ragged rows (width % 4 != 0), heights that end inside a 16-row tile, 1x1 taps
WS_CASES = [(3, 64, 128, 130, 190, 3, 1, 1, 1), (2, 80, 64, 257, 259, 3, 1, 1, 1), (2, 128, 192, 100, 132, 1, 1, 1, 0), (4, 256, 256, 64, 96, 3, 1, 1, 1),
            (2, 32, 32, 200, 262, 3, 1, 1, 1), (2, 64, 96, 130, 190, 3, 1, 1, 1), (4, 48, 32, 150, 170, 1, 1, 1, 0)]      # ... and its 32-output-channel variant
# shapes the wave-specialised TRANSPOSED kernel would take (convt_bf16x3_ws_kernel, built only with -DGC_CTWS=1: measured no faster than the
# one-role kernel; the cases stay as transposed-convolution coverage): up = 2, K % 16 == 0, N % 32 == 0, >= 192 workgroups -- its three
# tile shapes (64 oc x 16 x 16 | 64 oc x 8 x 32 | 32 oc x 16 x 32 q-pixels), several tiles per workgroup, odd chunk counts, ragged widths
TWS_CASES = [(4, 64, 128, 64, 64, 3, 2, 1, 2), (3, 48, 128, 100, 127, 3, 2, 1, 2), (3, 64, 32, 200, 130, 3, 2, 1, 2), (4, 32, 96, 70, 100, 3, 2, 1, 2),
             (4, 64, 64, 200, 260, 3, 2, 1, 2)]
# round 5: transposed 3x3 on small planes split over the input channels (convt_fused_bf16x3_kernel + splitk_finish_kernel): the networks' @8^2 / @16^2
# layers (8 and 3 slices), a ragged last slice, q-tiles that end inside the plane, the 16- and the 32-column tile
CT_SPLIT_CASES = [(4, 512, 512, 8, 8, 3, 2, 1, 2), (4, 512, 128, 16, 16, 3, 2, 1, 2), (2, 272, 64, 16, 12, 3, 2, 1, 2), (8, 512, 512, 8, 8, 3, 2, 1, 2), (1, 192, 96, 20, 9, 3, 2, 1, 2)]
# round 5: (2H + 1) x (2W + 1) transposed convolutions as an H x W main region + convt_edge_bf16x3_kernel (K % 16 == 0, K >= 256, N % 64 == 0, W % 32 == 0,
# H % 4 == 0, >= 512 workgroups in the main region): square and flat planes, row-pitched outputs (out_w >= 129), uneven channel quarters (17 chunks)
CT_EDGE_CASES = [(4, 256, 256, 64, 64, 3, 2, 1, 2), (8, 256, 256, 32, 64, 3, 2, 1, 2), (16, 272, 256, 32, 32, 3, 2, 1, 2), (4, 512, 64, 128, 128, 3, 2, 1, 2)]
# round 6: stride-2 3x3 convolutions on the wave-specialised E / O kernel (conv_s2ws_bf16x3_kernel: pad 0, K % 16 == 0, K >= 32, N % 64 == 0, >= 192 tiles x samples
# x oc blocks): both output-channel block widths (64 / 128), odd chunk counts, several tiles per workgroup, widths / heights that end inside a tile
S2WS_CASES = [(1, 32, 64, 513, 513, 3, 1, 2, 0), (3, 48, 128, 257, 259, 3, 1, 2, 0), (4, 32, 64, 205, 267, 3, 1, 2, 0), (4, 64, 256, 257, 257, 3, 1, 2, 0),
              (2, 80, 192, 223, 331, 3, 1, 2, 0)]
BF16_CASES = BF16_CASES + WS_CASES + TWS_CASES + SMALL_CASES + CT_SPLIT_CASES + CT_EDGE_CASES + S2WS_CASES + [(400, 64, 64, 1, 1, 3, 1, 1, 1)]      # ... and past the LDS: back on the general path


@pytest.mark.parametrize('case', BF16_CASES)
def test_conv2d_bf16x3_kernel(case, bf16x3_mode):
    """Split-bf16 MFMA path: ~5e-6 relative error per layer by construction; assert 5e-5 (parity bound is 1e-3)."""
    from gan_control_amd.models.op._backend import ConvGeom
    hip, emu = _be()
    b, K, N, h, w, k, up, down, pad = case
    gen = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(b, K, h, w, generator=gen)
    wt = torch.randn(k, k, K, N, generator=gen)
    si = torch.randn(b, K, generator=gen)
    so = torch.rand(b, N, generator=gen) + 0.5
    oh, ow = _out_size(h, k, up, down, pad, up > 1), _out_size(w, k, up, down, pad, up > 1)
    geom = ConvGeom(k, k, up, down, pad, pad, oh, ow)
    if case in WS_CASES:
        from gan_control_amd.utils.profiling import conv_variant
        assert conv_variant(geom, N, b, K, 'bf16x3', (h, w)).startswith('conv_bf16x3_ws_kernel'), 'this shape is meant to reach the wave-specialised kernel'
    if case in CT_EDGE_CASES:
        from gan_control_amd.utils.profiling import conv_variant
        assert conv_variant(geom, N, b, K, 'bf16x3', (h, w)).endswith('+edge|up2,down1,k3'), 'this shape is meant to run as main region + edge kernel'
    if case in CT_SPLIT_CASES:
        from gan_control_amd import _lib
        assert _lib.load().gc_conv2d_bf16x3_splitk_bytes(hip._desc(x, N, geom)) > 0, 'this shape is meant to be split over its input channels'
    if case in SMALL_CASES:
        from gan_control_amd.utils.profiling import conv_variant
        assert conv_variant(geom, N, b, K, 'bf16x3', (h, w)).startswith('conv_f32_small_kernel'), 'this shape is meant to reach the small-plane kernel (exact fp32 in this mode too)'
    if case in S2WS_CASES:
        from gan_control_amd.utils.profiling import conv_variant
        assert conv_variant(geom, N, b, K, 'bf16x3', (h, w)).startswith('conv_s2ws_bf16x3_kernel'), 'this shape is meant to reach the wave-specialised stride-2 kernel'
    tws = case in S2WS_CASES       # LDS-DMA weights behind counted waits: repeated next to other traffic below
    if case in TWS_CASES:
        from gan_control_amd.utils.profiling import conv_variant
        tws = conv_variant(geom, N, b, K, 'bf16x3', (h, w)).startswith('convt_bf16x3_ws_kernel')      # only in a library built with -DGC_CTWS=1
    for use_scales in (False, True):
        a = (si, so) if use_scales else (None, None)
        ref = emu.conv2d(x.double(), wt.double(), *[None if t is None else t.double() for t in a], geom)
        out = hip.conv2d(x.to(DEV), wt.to(DEV), *[None if t is None else t.to(DEV) for t in a], geom)
        assert rel_err(out, ref) < 5e-5, use_scales
        if tws:
            # LDS-DMA data reaches its readers by a counted vmcnt wait + a barrier: a misplaced wait passes whenever the DMA happens to land
            # first, so the launch is repeated next to other traffic and every output must be bit-identical to the first
            junk = torch.randn(16 << 20, device=DEV)
            for i in range(20):
                junk.mul_(1.0001)
                again = hip.conv2d(x.to(DEV), wt.to(DEV), *[None if t is None else t.to(DEV) for t in a], geom)
                assert torch.equal(again, out), ('run-to-run difference', i)
    if up == 1:
        dy = torch.randn(b, N, oh, ow, generator=gen)
        for use_scales in (False, True):
            a = (si, so) if use_scales else (None, None)
            ref = emu.conv2d_wgrad(x.double(), dy.double(), *[None if t is None else t.double() for t in a], geom)
            out = hip.conv2d_wgrad(x.to(DEV), dy.to(DEV), *[None if t is None else t.to(DEV) for t in a], geom)
            assert rel_err(out, ref) < 5e-5, ('wgrad', use_scales)


EPILOGUE_CASES = [c for c in BF16_CASES if c[1] >= 3 and c not in WS_CASES + SMALL_CASES + CT_SPLIT_CASES + CT_EDGE_CASES + S2WS_CASES][::2] + WS_CASES + SMALL_CASES[1:4] + SMALL_UP_CASES[:2] + SMALL_GROUP_CASES[:3] + CT_SPLIT_CASES[:3] + CT_EDGE_CASES[:3] + S2WS_CASES[:4]


@pytest.mark.parametrize('mode', ['f32', 'bf16x3'])
@pytest.mark.parametrize('case', EPILOGUE_CASES)
def test_conv2d_fused_epilogue(case, mode):
    """gc_conv2d_fused_*: (noise +) bias + leaky-ReLU in the convolution epilogue == convolution followed by gc_bias_act_f32, bit for bit."""
    from gan_control_amd.models.op._backend import ConvGeom
    hip, emu = _be()
    prev, hip.conv_mode = hip.conv_mode, mode
    try:
        b, K, N, h, w, k, up, down, pad = case
        gen = torch.Generator().manual_seed(hash(case) & 0xFFF)
        x = torch.randn(b, K, h, w, generator=gen).to(DEV)
        wt = torch.randn(k, k, K, N, generator=gen).to(DEV)
        si, so = torch.randn(b, K, generator=gen).to(DEV), (torch.rand(b, N, generator=gen) + 0.5).to(DEV)
        oh, ow = _out_size(h, k, up, down, pad, up > 1), _out_size(w, k, up, down, pad, up > 1)
        geom = ConvGeom(k, k, up, down, pad, pad, oh, ow)
        bias = torch.randn(N, generator=gen).to(DEV)
        nz, nw = torch.randn(b, 1, oh, ow, generator=gen).to(DEV), torch.randn(1, generator=gen).to(DEV)
        for scales in ((None, None), (si, so)):
            plain = hip.conv2d(x, wt, *scales, geom).contiguous()      # (a transposed convolution may hand out row-pitched views)
            fused = hip.conv2d(x, wt, *scales, geom, epilogue=(bias, None, None, 0.2, 2 ** 0.5, True))
            assert torch.equal(fused, hip.bias_act(plain, bias, None, None, 0.2, 2 ** 0.5))
            fused = hip.conv2d(x, wt, *scales, geom, epilogue=(bias, nz, nw, 0.2, 2 ** 0.5, True))
            assert torch.equal(fused, hip.bias_act(plain, bias, nz, nw, 0.2, 2 ** 0.5))
            fused = hip.conv2d(x, wt, *scales, geom, epilogue=(bias, None, None, 1.0, 1.0, False))
            assert torch.equal(fused, plain + bias.reshape(1, -1, 1, 1))
            fused = hip.conv2d(x, wt, *scales, geom, epilogue=(None, None, None, 0.2, 2 ** 0.5, True))
            assert torch.equal(fused, hip.bias_act(plain, None, None, None, 0.2, 2 ** 0.5))
            res = torch.randn(plain.shape, generator=gen).to(DEV)
            fused = hip.conv2d(x, wt, *scales, geom, epilogue=(None, None, None, 1.0, 1.0, False, res))
            assert torch.equal(fused, plain + res)
            fused = hip.conv2d(x, wt, *scales, geom, epilogue=(bias, nz, nw, 0.2, 2 ** 0.5, True, res))
            assert torch.equal(fused, hip.bias_act(plain, bias, nz, nw, 0.2, 2 ** 0.5) + res)
        # against the independent emulation in fp64
        ref = emu.conv2d(x.cpu().double(), wt.cpu().double(), si.cpu().double(), so.cpu().double(), geom,
                         epilogue=(bias.cpu().double(), nz.cpu().double(), nw.cpu().double(), 0.2, 2 ** 0.5, True))
        out = hip.conv2d(x, wt, si, so, geom, epilogue=(bias, nz, nw, 0.2, 2 ** 0.5, True))
        assert rel_err(out, ref) < (5e-6 if mode == 'f32' else 5e-5)
    finally:
        hip.conv_mode = prev


def test_conv2d_bias_act_autograd_matches_two_pass():
    """conv2d_bias_act (one launch) and conv2d -> fused_leaky_relu (two) give the same values and first / second-order gradients."""
    from gan_control_amd.models.op import conv2d_gradfix, fused_leaky_relu
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(2, 24, 33, 40, generator=gen).to(DEV).requires_grad_(True)
    w = torch.randn(40, 24, 3, 3, generator=gen).to(DEV).requires_grad_(True)
    b = torch.randn(40, generator=gen).to(DEV).requires_grad_(True)
    for stride, pad in [(1, 1), (2, 0)]:
        outs = []
        for fused in (True, False):
            if fused:
                y = conv2d_gradfix.conv2d_bias_act(x, w, b, stride=stride, padding=pad, weight_scale=0.1)
            else:
                y = fused_leaky_relu(conv2d_gradfix.conv2d(x, w, stride=stride, padding=pad, weight_scale=0.1), b)
            g1 = torch.autograd.grad(y.square().mean(), [x, w, b], retain_graph=True)
            gx, = torch.autograd.grad(y.sum() + (y * y).sum(), x, create_graph=True)
            g2 = torch.autograd.grad(gx.pow(2).sum(), [x, w, b])
            outs.append([y.detach(), gx.detach(), *g1, *g2])
        for a, c in zip(*outs):
            assert rel_err(a, c) < 1e-5


@pytest.mark.parametrize('size', [64, 256, 512, 1024])
def test_network_golden_bf16x3(size, bf16x3_mode):
    oc.check_network(size, DEV)


def test_step_golden_bf16x3(bf16x3_mode):
    import step_checks
    step_checks.check_step(DEV, param_tol=1e-2)


@pytest.mark.parametrize('mode', ['f32', 'bf16x3'])
@pytest.mark.parametrize('name', ['step_512', 'step_1024'])
def test_step_golden_baseline_sizes(name, mode):
    """The full iteration (D step, R1, G step, path length, EMA) at the BASELINE resolutions -- 512x512 batch 4 and 1024x1024
    batch 2 (the largest the 64 GiB build container could run the reference at) -- against the reference-driven fixtures: loss
    scalars, path lengths, global and per-parameter gradient norms of all four backward passes, parameters after Adam."""
    import step_checks
    hip, _ = _be()
    prev, hip.conv_mode = hip.conv_mode, mode
    try:
        # losses and global gradient norms: 2e-3 in both modes.  Per-parameter gradient norms: 2e-3 in exact fp32, 1e-2 in split-bf16
        # (16 mantissa bits per product; the worst parameter's gradient cancels to ~1/700 of its terms: measured up to 4.3e-3)
        step_checks.check_step(DEV, tol=2e-3, name=name, param_tol=None if mode == 'f32' else 1e-2)
    finally:
        hip.conv_mode = prev
        torch.cuda.empty_cache()


@pytest.mark.parametrize('mode', ['f32', 'bf16x3'])
def test_headline_iteration_against_the_reference(mode):
    """The workload bench.py times -- 1024 x 1024, 4 images per GPU (one full minibatch-stddev group) -- with the trainer built EXACTLY as
    bench.py builds it (fused Adam, weight cache with batched refill, grouped style path, per-sample weight-gradient route, every fusion at its
    default), against the reference's own modules and trainer maths (tests/golden/step_1024_b4.npz, oracle/make_golden.py::golden_step_isolated):
    each of the four backward passes from the procedural weights -- losses, path lengths, which parameters get a gradient, global and
    per-parameter gradient norms and 64 sampled gradient ELEMENTS per parameter tensor (direction, not just length).
    Tolerances = 2 x what tools/headline_parity_probe.py measured (profiles/headline_parity_r04.json)."""
    import step_checks
    from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config
    hip, _ = _be()
    prev, hip.conv_mode = hip.conv_mode, mode
    try:
        # measured (profiles/headline_parity_r04.json): losses <= 4e-4; per-parameter norms 1.0e-3 (f32) / 2.1e-3 (bf16x3); sampled elements 9.4e-4 / 2.2e-3 in the
        # plain passes and 4.6e-3 / 9.3e-3 in the path-length pass (elements are bounded at 2 x param_tol, double-backward passes at 3 x that)
        step_checks.check_isolated(DEV, name='step_1024_b4', tol=2e-3, param_tol=None if mode == 'f32' else 5e-3,
                                   trainer=lambda size, batch: GeneratorTrainer(default_config(size, batch), device=DEV, seed=0),
                                   ratchet=step_checks.load_measured().get('%s/' % 'step_1024_b4' + mode))
    finally:
        hip.conv_mode = prev
        torch.cuda.empty_cache()


def test_headline_iteration_bf16x3_vs_f32():
    """The bench workload itself (1024x1024, 4 images: no CPU fixture fits the build container): each of the four backward passes of
    the iteration in split-bf16 arithmetic against the same pass in exact fp32 on the same inputs and weights -- losses, path lengths
    and the gradient norm of every parameter.  Every pass starts from the un-updated weights (see step_checks.check_step)."""
    import step_checks
    from gan_control_amd.trainers.utils import requires_grad
    hip, _ = _be()
    gen = torch.Generator().manual_seed(99)
    real = (torch.rand(4, 3, 1024, 1024, generator=gen) * 2 - 1).to(DEV)
    z_d, z_g, z_pl = (torch.randn(n, 512, generator=gen).to(DEV) for n in (4, 4, 2))
    pl_noise = torch.randn(2, 3, 1024, 1024, generator=gen).to(DEV)
    runs = {}
    prev = hip.conv_mode
    try:
        tr = step_checks.make_trainer(DEV, size=1024, batch=4)
        fresh_g = {k: v.detach().clone() for k, v in tr.generator.state_dict().items()}
        fresh_d = {k: v.detach().clone() for k, v in tr.discriminator.state_dict().items()}
        grads = lambda m: {n: p.grad.norm().clone() for n, p in m.named_parameters() if p.grad is not None}
        for mode in ('f32', 'bf16x3'):
            hip.conv_mode = mode
            rec = {}
            for phase in ('d', 'r1', 'g', 'pl'):
                tr.generator.load_state_dict(fresh_g)
                tr.discriminator.load_state_dict(fresh_d)
                tr.mean_path_length = 0
                on_d = phase in ('d', 'r1')
                requires_grad(tr.generator, not on_d); requires_grad(tr.discriminator, on_d)
                if phase == 'd':
                    tr.discriminator_step([[z_d]], [real], noise=oc.seeded_noise(1024, 4, 1, DEV))
                elif phase == 'r1':
                    tr.discriminator_regularize_step([real])
                elif phase == 'g':
                    tr.generator_step([[z_g]], noise=oc.seeded_noise(1024, 4, 2, DEV))
                else:
                    tr.generator_regularize_step(noise=oc.seeded_noise(1024, 2, 3, DEV), pl_noise=pl_noise, z=[z_pl])
                rec[phase] = grads(tr.discriminator if on_d else tr.generator)
            rec['stats'] = {k: tr.stats[k].clone() if torch.is_tensor(tr.stats[k]) else tr.stats[k] for k in tr.stats}
            runs[mode] = rec
    finally:
        hip.conv_mode = prev
        del tr
        torch.cuda.empty_cache()
    a, b = runs['bf16x3'], runs['f32']
    for k in ('d_loss', 'd_r1_loss', 'g_adv_loss', 'g_path_loss'):
        assert abs(float(a['stats'][k]) - float(b['stats'][k])) <= 1e-3 * max(1e-3 if k == 'd_r1_loss' else 1.0, abs(float(b['stats'][k]))), k
    assert rel_err(a['stats']['path_lengths'], b['stats']['path_lengths']) <= 1e-3
    for phase, tol in (('d', 2e-3), ('r1', 6e-3), ('g', 2e-3), ('pl', 6e-3)):
        assert a[phase].keys() == b[phase].keys()
        total = float(torch.stack(list(b[phase].values())).norm())
        for n in b[phase]:
            if n.endswith('noise.weight'):
                continue            # cancelling scalar sums: compared as a group in step_checks
            ref = float(b[phase][n])
            assert abs(float(a[phase][n]) - ref) <= tol * max(ref, 1e-3 * total), (phase, n, float(a[phase][n]), ref)


def test_weight_cache_survives_fused_optimizer_and_data_writes_gpu():
    """Fused Adam (the device default) and the EMA's ``.data`` writes do not bump version counters: the cache of derived weight forms
    must be dropped by the optimiser hook / accumulate (regression: training ran on the first iteration's convolution weights)."""
    from test_ops_emulated import _fused_step_conv_check
    _fused_step_conv_check(DEV)


@pytest.mark.parametrize('mode', ['bf16x3', 'f32'])
def test_batched_weight_forms_are_bit_identical_gpu(mode):
    """Three iterations with the derived weight forms of a whole network refilled by the grouped kernels (gc_weight_layout_grouped_f32,
    gc_conv2d_pack_weights_bf16x3_grouped), one form at a time, and without any cache: the same parameters, bit for bit."""
    from gan_control_amd.models.op import weight_cache
    from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config
    hip, _ = _be()
    prev_mode, hip.conv_mode = hip.conv_mode, mode
    out = []
    try:
        for enabled, batched in ((True, True), (True, False), (False, False)):
            weight_cache.clear()
            weight_cache.ENABLED, prev = enabled, weight_cache.ENABLED
            weight_cache.BATCHED, prev_b = batched, weight_cache.BATCHED
            before = weight_cache.stats['batched']
            try:
                tr = GeneratorTrainer(default_config(64, 4), device=DEV, seed=0)
                real = tr.synthetic_batch()
                for i in range(3):
                    tr.train_iteration(i, real)
                out.append({k: v.clone() for k, v in list(tr.generator.state_dict().items()) + list(tr.discriminator.state_dict().items())})
                assert (weight_cache.stats['batched'] > before) == batched
            finally:
                weight_cache.ENABLED, weight_cache.BATCHED = prev, prev_b
    finally:
        hip.conv_mode = prev_mode
    for k in out[0]:
        assert torch.equal(out[0][k], out[1][k]) and torch.equal(out[0][k], out[2][k]), k
    weight_cache.clear()


def _pitched(t, pitch):
    """A row-pitched copy of a dense [B, C, H, W] tensor: the layout gc_conv_desc.out_pitch / gc_upfirdn2d_pitched_f32 write, with
    garbage in the padding columns (nothing may read them)."""
    b, c, h, w = t.shape
    buf = torch.full((b, c, h, pitch), float('nan'), device=t.device, dtype=t.dtype)
    buf[..., :w] = t
    return buf[..., :w]


@pytest.mark.parametrize('shape', [(2, 5, 129, 131), (1, 3, 257, 257), (3, 2, 70, 161)])
def test_row_pitched_fir_and_plane_dot(shape):
    """Row-pitched tensors (the aligned-row layout of the (2H + 1)- / (H + 1)-wide intermediates): the Blur tile kernel reads and writes
    them, the plane reductions read them -- bit-identical to the dense path."""
    from gan_control_amd import _lib
    hip, _ = _be()
    prev, hip.conv_mode = hip.conv_mode, 'bf16x3'
    try:
        gen = torch.Generator().manual_seed(4)
        x = torch.randn(shape, generator=gen).to(DEV)
        k4 = torch.rand(4, 4, generator=gen).to(DEV)
        b, c, h, w = shape
        xp = _pitched(x, (w + 31) // 32 * 32 + 32)
        assert _lib.row_pitch(xp) and not _lib.row_pitch(x)
        for pad, (oh, ow) in ((1, (h - 1, w - 1)), (2, (h + 1, w + 1))):
            dense = hip.upfirdn2d(x, k4, 1, 1, pad, pad, oh, ow, True).contiguous()
            assert torch.equal(hip.upfirdn2d(xp, k4, 1, 1, pad, pad, oh, ow, True), dense), (pad, 'pitched input')
            bias, nz, nw = torch.randn(c, generator=gen).to(DEV), torch.randn(b, 1, oh, ow, generator=gen).to(DEV), torch.randn(1, generator=gen).to(DEV)
            assert torch.equal(hip.upfirdn2d_act(xp, k4, pad, pad, oh, ow, True, bias, nz, nw, 0.2, 1.4), hip.upfirdn2d_act(x, k4, pad, pad, oh, ow, True, bias, nz, nw, 0.2, 1.4))
            if ow % 4 and ow >= 129:          # an odd-width output comes back row-pitched itself
                assert _lib.row_pitch(hip.upfirdn2d(x, k4, 1, 1, pad, pad, oh, ow, True))
        y = torch.randn(shape, generator=gen).to(DEV)
        den = (torch.rand(b, c, generator=gen) + 0.5).to(DEV)
        ref = (x.double() * y.double()).sum((2, 3))
        p32 = (w + 31) // 32 * 32
        for a_, b_ in ((xp, y), (x, _pitched(y, p32)), (xp, _pitched(y, p32 + 64))):
            assert rel_err(hip.plane_dot(a_, b_), ref) < 1e-5
            assert rel_err(hip.plane_dot(a_, b_, den), ref / den.double()) < 1e-5
        # only the layout the kernels write themselves is read in place (rows a multiple of 32 floats apart, 16-byte aligned): a column
        # slice of a user tensor has the same stride pattern and must be treated as an ordinary non-contiguous tensor
        assert not _lib.row_pitch(_pitched(y, w + 7)) and not _lib.row_pitch(torch.randn(b, c, h, p32 + 32, device=DEV)[..., 1:w + 1])
        from gan_control_amd.models.op import upfirdn2d, conv2d_gradfix
        sliced = torch.randn(b, c, h, p32 + 32, device=DEV)[..., 1:w + 1]
        assert torch.equal(upfirdn2d(sliced, k4, pad=(1, 1)), upfirdn2d(sliced.contiguous(), k4, pad=(1, 1)))
        # public results are dense tensors (the reference's callers use .view()): the row-pitched layout stays inside the package
        if (w + 1) % 4 and w + 1 >= 129:
            pub = upfirdn2d(x, k4, pad=(2, 2))
            assert pub.is_contiguous() and pub.view(-1).numel() == pub.numel()
            assert _lib.row_pitch(upfirdn2d(x, k4, pad=(2, 2), _internal=True))
            wt = torch.randn(c, 64, 3, 3, generator=gen).to(DEV)
            if c >= 16:
                assert conv2d_gradfix.conv_transpose2d(x, wt, stride=2).is_contiguous()
    finally:
        hip.conv_mode = prev


@pytest.mark.parametrize('case', [(2, 32, 64, 131, 133, 3), (1, 64, 64, 257, 257, 3), (2, 48, 96, 129, 161, 1), (4, 32, 64, 257, 257, 3), (3, 32, 128, 257, 261, 3), (2, 64, 128, 131, 197, 3), (2, 96, 256, 67, 131, 1)])
def test_stride2_kernels_read_row_pitched_input(case, bf16x3_mode):
    """The split-bf16 stride-2 convolution and its weight gradient on a row-pitched input (gc_conv_desc.in_pitch): bit-identical to the dense
    input; the transposed convolution's pitched output (gc_conv_desc.out_pitch) equals its dense values."""
    from gan_control_amd.models.op._backend import ConvGeom
    from gan_control_amd import _lib
    hip, _ = _be()
    b, K, N, h, w, k = case
    gen = torch.Generator().manual_seed(8)
    x = torch.randn(b, K, h, w, generator=gen).to(DEV)
    wt = torch.randn(k, k, K, N, generator=gen).to(DEV)
    si, so = torch.randn(b, K, generator=gen).to(DEV), (torch.rand(b, N, generator=gen) + 0.5).to(DEV)
    oh, ow = (h - k) // 2 + 1, (w - k) // 2 + 1
    geom = ConvGeom(k, k, 1, 2, 0, 0, oh, ow)
    xp = _pitched(x, (w + 31) // 32 * 32)
    assert torch.equal(hip.conv2d(xp, wt, si, so, geom), hip.conv2d(x, wt, si, so, geom))
    dy = torch.randn(b, N, oh, ow, generator=gen).to(DEV)
    assert torch.equal(hip.conv2d_wgrad(xp, dy, si, so, geom), hip.conv2d_wgrad(x, dy, si, so, geom))
    if k == 3:
        # the adjoint geometry (input gradient of the stride-2 convolution = a transposed convolution): (2 oh + 1)-wide rows come back pitched
        tg = ConvGeom(3, 3, 2, 1, 2, 2, 2 * oh + 1, 2 * ow + 1)
        wa = torch.randn(3, 3, N, K, generator=gen).to(DEV)
        out = hip.conv2d(dy, wa, so, si, tg)
        import gan_control_amd.models.op._backend as be_mod
        be_mod._PITCHED_OUTPUT, keep = False, be_mod._PITCHED_OUTPUT
        try:
            dense = hip.conv2d(dy, wa, so, si, tg)
        finally:
            be_mod._PITCHED_OUTPUT = keep
        # (a transposed layer on a small plane is split over its input channels only in the dense form -- the finish pass writes dense rows --, and the
        #  two forms then add the channel slices in different orders: equal to rounding there, bit for bit everywhere else)
        split = _lib.load().gc_conv2d_bf16x3_splitk_bytes(hip._desc(dy, K, tg)) > 0
        assert dense.is_contiguous() and (rel_err(out, dense) < 1e-5 if split else torch.equal(out, dense))
        if (2 * ow + 1) >= 129:
            assert _lib.row_pitch(out) % 32 == 0 and _lib.row_pitch(out) >= 2 * ow + 1


def test_split_fc_gpu():
    oc.check_split_fc(DEV)


def test_mixing_truncation_and_stored_noise_gpu():
    oc.check_mixing_truncation(DEV)


def test_noise_modes_zeros_and_id_zeros_gpu():
    oc.check_noise_modes(DEV)


def test_transfer_learning_load_gpu():
    oc.check_transfer_learning(DEV)


def test_misc_modules_gpu():
    oc.check_misc(DEV)


@pytest.mark.parametrize('name,size,batch', [('ffhq', 64, 8), ('metfaces', 64, 8), ('afhq', 64, 8)])
def test_trainer_from_shipped_config_gpu(name, size, batch):
    """A trainer built from the hot-path fields of each shipped configuration (split_fc mapping network; ADA on for metfaces /
    afhq) runs a full iteration on the HIP kernels."""
    oc.check_config_ingestion(DEV, name, size=size, batch=batch)


def test_ddp_buckets_launch_from_hooks():
    """RCCL path (one rank, collective forced on): once a phase has run, every gradient bucket of D step, R1, G step and the
    path-length step is launched from a gradient hook DURING backward -- none is left for finish() -- and the gradients the
    optimiser reads are views of the bucket buffers (no copy back)."""
    import json
    import os
    import subprocess
    import sys
    from conftest import REPO
    env = dict(os.environ, GANCONTROL_FORCE_DDP='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', '29541', os.path.join(REPO, 'tests', 'ddp_probe.py')]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith('PROBE ')][-1][6:])
    assert rec['grads_are_bucket_views']
    for net, phases in (('d', ('d', 'r1')), ('g', ('g', 'pl'))):
        for ph in phases:
            assert rec['first'][net][ph]['hook'] == 0, 'the first occurrence of a phase learns its gradient set'
            last = rec['last'][net][ph]
            assert last['finish'] == 0 and last['late'] == 0 and last['hook'] >= 1, (net, ph, last)
    assert all(v == v for v in rec['losses'].values())


def test_bench_ddp_path_single_rank():
    """bench.py through torch.distributed.run with one rank and the collective path forced on: RCCL init, bucketed
    gradient all-reduce from autograd hooks, barrier/MAX timing -- everything the multi-GPU launch does."""
    import json
    import os
    import subprocess
    import sys
    from conftest import REPO
    env = dict(os.environ, GANCONTROL_FORCE_DDP='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', '29533', os.path.join(REPO, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
           '--size', '64', '--batch-per-gpu', '4', '--no-cpu-baseline']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    rec = json.loads(line)
    assert rec['n_gpus'] == 1 and rec['value'] > 0 and rec['unit'] == 'images/sec'
    assert all(v == v for v in rec['losses'].values()), 'NaN loss'


def test_transposed_conv_grads_bf16x3(bf16x3_mode):
    """conv_transpose2d at >= 64 channels: forward on the up=2 kernel, input gradient on the down=2 kernel, weight
    gradient on the stride-2 split-bf16 wgrad with swapped operands."""
    import torch.nn.functional as F
    from torch import autograd
    from gan_control_amd.models.op import conv2d_gradfix
    gen = torch.Generator().manual_seed(23)
    for (b, ic, oc, h, w) in [(2, 64, 64, 33, 40), (1, 128, 64, 20, 37)]:
        x = torch.randn(b, ic, h, w, generator=gen)
        wt = torch.randn(ic, oc, 3, 3, generator=gen)
        xr, wr = x.double().requires_grad_(True), wt.double().requires_grad_(True)
        xp, wp = x.to(DEV).requires_grad_(True), wt.to(DEV).requires_grad_(True)
        ref = F.conv_transpose2d(xr, wr, stride=2)
        out = conv2d_gradfix.conv_transpose2d(xp, wp, stride=2)
        assert rel_err(out, ref) < 5e-5
        go = torch.randn_like(ref)
        gr = autograd.grad(ref, [xr, wr], go)
        gp = autograd.grad(out, [xp, wp], go.float().to(DEV))
        for a, c in zip(gp, gr):
            assert rel_err(a, c) < 5e-5


@pytest.mark.parametrize('shape,noise,self_dot', [((2, 5, 33, 40), True, True), ((2, 5, 33, 40), False, True), ((3, 4, 150, 130), True, True),
                                                  ((2, 6, 20, 17), True, False), ((1, 3, 7), False, False), ((2, 4, 129, 129), False, True)])
def test_bias_act_bwd_reduce_adjoint(shape, noise, self_dot):
    """The fused second-order pass (gc_bias_act_bwd_reduce_adjoint_f32) against the ATen formulas it replaces, through autograd:
    the same double-backward computed with grad mode on (ATen branch) and off (HIP kernel)."""
    from gan_control_amd.models.op import fused_act
    gen = torch.Generator().manual_seed(sum(shape) + noise + 2 * self_dot)
    slope, gain = 0.2, 2 ** 0.5
    y = torch.randn(*shape, generator=gen).to(DEV)
    y = torch.where(y.abs() < 0.05, torch.full_like(y, 0.3), y).requires_grad_(True)
    gy = torch.randn(*shape, generator=gen).to(DEV).requires_grad_(True)
    nz = torch.randn(shape[0], 1, *shape[2:], generator=gen).to(DEV) if noise else None
    bias = torch.randn(shape[1], generator=gen).to(DEV).requires_grad_(True) if self_dot else None
    nw = torch.randn(1, generator=gen).to(DEV).requires_grad_(True) if (self_dot and noise) else None
    outs = fused_act._BiasActGradReduce.apply(gy, y, nz, slope, gain, bias, nw, self_dot)
    live = [o for o in outs if o.numel() > 0 and o.requires_grad]
    cots = [torch.randn(o.shape, generator=gen).to(DEV) for o in live]
    wrt = [t for t in (gy, y, bias, nw) if t is not None]
    ref = torch.autograd.grad(live, wrt, cots, retain_graph=True, create_graph=True, allow_unused=True)      # grad mode on: ATen formulas
    out = torch.autograd.grad(live, wrt, cots, retain_graph=True, allow_unused=True)                          # grad mode off: the HIP pass
    for r, o, name in zip(ref, out, ('gy', 'y', 'bias', 'noise_w')):
        assert (r is None) == (o is None), name
        if r is not None:
            assert rel_err(o, r.detach()) < 1e-5, name


@pytest.mark.parametrize('up,down', [(2, 1), (1, 2), (1, 1)])
@pytest.mark.parametrize('shape', [(2, 3, 70, 131), (1, 3, 16, 40), (1, 2, 129, 65)])
def test_upfirdn2d_12x12_tile_kernel(up, down, shape):
    """The LDS-tiled 12 x 12 kernel (firK_tile_kernel: the sym6 x sym6 anti-aliasing passes of ADA and their adjoints) across tile
    edges, pads of both signs and both tap orders, against the emulated formula in fp64."""
    hip, emu = _be()
    gen = torch.Generator().manual_seed(sum(shape) + 10 * up + down)
    x = torch.randn(*shape, generator=gen)
    k = torch.randn(12, 12, generator=gen)
    for p0, p1 in [(0, 0), (6, 5), (7, 6), (11, 1), (-3, -3), (2, 9)]:
        oh, ow = (shape[2] * up + p0 + p1 - 12) // down + 1, (shape[3] * up + p0 + p1 - 12) // down + 1
        if oh < 8 or ow < 32:
            continue
        for flip in (True, False):
            ref = emu.upfirdn2d(x.double(), k.double(), up, down, p0, p0, oh, ow, flip)
            out = hip.upfirdn2d(x.to(DEV), k.to(DEV), up, down, p0, p0, oh, ow, flip)
            assert out.shape == ref.shape
            assert rel_err(out, ref) < 2e-6, (p0, p1, flip)


@pytest.mark.parametrize('shape,out_hw', [((2, 3, 40, 56), (40, 56)), ((3, 3, 64, 64), (50, 70)), ((1, 1, 7, 9), (20, 5))])
def test_affine_warp_kernel(shape, out_hw):
    """gc_affine_warp_bilinear_f32 and its adjoint against F.grid_sample(bilinear, zeros, align_corners=False) (fp64): rotations,
    scales and shifts that send part of the output outside the image."""
    import math
    hip, emu = _be()
    gen = torch.Generator().manual_seed(sum(shape))
    b, c, h, w = shape
    x = torch.randn(*shape, generator=gen)
    th = torch.rand(b, generator=gen) * 2 * math.pi
    sc = torch.rand(b, generator=gen) + 0.5
    mat = torch.stack([sc * torch.cos(th), -sc * torch.sin(th), torch.rand(b, generator=gen) * w * 0.5,
                       sc * torch.sin(th), sc * torch.cos(th), torch.rand(b, generator=gen) * h * 0.5 - 3], dim=1).float()
    ref = emu.affine_warp(x.double(), mat.double(), h, w, out_hw[0], out_hw[1], False)
    out = hip.affine_warp(x.to(DEV), mat.to(DEV), h, w, out_hw[0], out_hw[1], False)
    assert rel_err(out, ref) < 1e-5
    g = torch.randn(b, c, *out_hw, generator=gen)
    ref = emu.affine_warp(g.double(), mat.double(), h, w, out_hw[0], out_hw[1], True)
    out = hip.affine_warp(g.to(DEV), mat.to(DEV), h, w, out_hw[0], out_hw[1], True)
    assert rel_err(out, ref) < 1e-5
    # autograd: the adjoint is the gradient, the forward is the gradient of the adjoint
    from gan_control_amd.models.op import affine_warp_bilinear
    xp = x.to(DEV).requires_grad_(True)
    y = affine_warp_bilinear(xp, mat.to(DEV), *out_hw)
    gp = g.to(DEV).requires_grad_(True)
    gx, = torch.autograd.grad(y, xp, gp, create_graph=True)
    assert rel_err(gx, ref) < 1e-5
    v = torch.randn(*shape, generator=gen).to(DEV)
    gg, = torch.autograd.grad((gx * v).sum(), gp)
    assert rel_err(gg, emu.affine_warp(v.cpu().double(), mat.double(), h, w, out_hw[0], out_hw[1], False)) < 1e-5


@pytest.mark.parametrize('shape,pads', [((2, 3, 20, 30), (5, 7, 3, 9)), ((1, 2, 8, 8), (7, 7, 7, 7)), ((3, 1, 33, 17), (0, 4, 2, 0)), ((1, 1, 5, 6), (0, 0, 0, 0))])
def test_reflect_pad_kernel(shape, pads):
    import torch.nn.functional as F
    from gan_control_amd.models.op import reflect_pad
    gen = torch.Generator().manual_seed(sum(shape) + sum(pads))
    x = torch.randn(*shape, generator=gen)
    xr = x.double().requires_grad_(True)
    xp = x.to(DEV).requires_grad_(True)
    ref = F.pad(xr, pads, mode='reflect')
    out = reflect_pad(xp, pads)
    assert torch.equal(out.cpu().double(), ref.detach())
    g = torch.randn(ref.shape, generator=gen)
    gr, = torch.autograd.grad(ref, xr, g.double())
    gp, = torch.autograd.grad(out, xp, g.to(DEV))
    assert rel_err(gp, gr) < 1e-6
    with pytest.raises(ValueError):
        reflect_pad(xp, (shape[3], 0, 0, 0))


@pytest.mark.parametrize('case', BF16_CASES[::3])
def test_conv2d_bf16_kernel(case):
    """Plain bf16 arithmetic (one MFMA per product on bf16-rounded operands, fp32 accumulate / storage): ~3e-3 relative error per layer by
    construction; asserted at 2e-2.  Not a parity mode: BASELINE config[1]'s precision."""
    from gan_control_amd.models.op._backend import ConvGeom
    hip, emu = _be()
    prev, hip.conv_mode = hip.conv_mode, 'bf16'
    try:
        b, K, N, h, w, k, up, down, pad = case
        gen = torch.Generator().manual_seed(hash(case) & 0xFFFF)
        x = torch.randn(b, K, h, w, generator=gen)
        wt = torch.randn(k, k, K, N, generator=gen)
        si = torch.randn(b, K, generator=gen)
        so = torch.rand(b, N, generator=gen) + 0.5
        oh, ow = _out_size(h, k, up, down, pad, up > 1), _out_size(w, k, up, down, pad, up > 1)
        geom = ConvGeom(k, k, up, down, pad, pad, oh, ow)
        ref = emu.conv2d(x.double(), wt.double(), si.double(), so.double(), geom)
        out = hip.conv2d(x.to(DEV), wt.to(DEV), si.to(DEV), so.to(DEV), geom)
        assert rel_err(out, ref) < 2e-2
        # the split-bf16 mode on the same call is three orders closer: the two builds really are different kernels
        hip.conv_mode = 'bf16x3'
        fine = hip.conv2d(x.to(DEV), wt.to(DEV), si.to(DEV), so.to(DEV), geom)
        hip.conv_mode = 'bf16'
        if K >= 16 and -(-ow // up) > 8 and not (k == 1 and (K <= 4 or N <= 4)):
            assert rel_err(fine, ref) < 5e-5 < rel_err(out, ref)
        if up == 1:
            dy = torch.randn(b, N, oh, ow, generator=gen)
            ref = emu.conv2d_wgrad(x.double(), dy.double(), si.double(), so.double(), geom)
            out = hip.conv2d_wgrad(x.to(DEV), dy.to(DEV), si.to(DEV), so.to(DEV), geom)
            assert rel_err(out, ref) < 2e-2
    finally:
        hip.conv_mode = prev


def test_bf16_mode_network_and_iteration():
    """--precision bf16: G / D forward within 5e-2 of the reference image at 64x64 (looser check of its own) and one
    full training iteration with finite losses that stay near the exact-fp32 iteration's."""
    import step_checks
    from gan_control_amd.trainers.utils import requires_grad
    hip, _ = _be()
    prev, hip.conv_mode = hip.conv_mode, 'bf16'
    try:
        oc.check_network(64, DEV, tol=5e-2)
        stats = {}
        for mode in ('f32', 'bf16'):
            hip.conv_mode = mode
            tr = step_checks.make_trainer(DEV, size=64, batch=4)
            torch.manual_seed(1)
            real = tr.synthetic_batch()
            tr.gen.manual_seed(5)
            tr.train_iteration(0, real)
            stats[mode] = {k: float(v) for k, v in tr.stats.items() if torch.is_tensor(v) and v.numel() == 1}
        for k in ('d_loss', 'g_adv_loss', 'g_path_loss'):
            a, b = stats['bf16'][k], stats['f32'][k]
            assert a == a and abs(a - b) <= 0.1 * max(1.0, abs(b)), (k, a, b)
    finally:
        hip.conv_mode = prev


SMALL_GEMM_CASES = [
    # (M, K, N, a transposed view, b layout 'kn' (n contiguous) / 'nk' (k contiguous), bias): inner extent <= 8
    (512, 4, 512, True, 'kn', False), (256, 8, 512, True, 'kn', False),       # EqualLinear weight gradients g^T @ x
    (32, 8, 512, True, 'kn', False), (64, 3, 100, False, 'kn', True), (4, 4, 4, False, 'kn', True), (70, 1, 33, False, 'nk', True), (3, 2, 1, True, 'nk', False),
]


@pytest.mark.parametrize('case', SMALL_GEMM_CASES)
def test_small_gemm(case):
    """gc_small_gemm_f32 == alpha * (a @ b) + beta * bias in fp64, through transposed views (strides, no copies)."""
    hip, _ = _be()
    m, k, n, a_t, b_layout, with_bias = case
    gen = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    a = (torch.randn(k, m, generator=gen).to(DEV).t() if a_t else torch.randn(m, k, generator=gen).to(DEV))
    b = (torch.randn(n, k, generator=gen).to(DEV).t() if b_layout == 'nk' else torch.randn(k, n, generator=gen).to(DEV))
    bias = torch.randn(n, generator=gen).to(DEV) if with_bias else None
    assert hip.small_gemm_ok(a, b)
    out = hip.small_gemm(a, b, bias, 0.01, 0.37)
    ref = 0.37 * (a.double() @ b.double()) + (0.01 * bias.double() if with_bias else 0.0)
    assert out.shape == (m, n) and rel_err(out, ref) < 2e-6
    assert torch.equal(out, hip.small_gemm(a, b, bias, 0.01, 0.37))          # deterministic
    assert not hip.small_gemm_ok(torch.randn(4, 512, device=DEV), torch.randn(512, 512, device=DEV))        # skinny products stay on the GEMM library


def test_equal_linear_small_gemm_autograd():
    """equal_linear / scaled_mm on the small-product kernels: values, first and second derivatives against plain fp64 ATen."""
    from gan_control_amd.models.op.linear import equal_linear, scaled_mm
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(4, 512, generator=gen).to(DEV).requires_grad_(True)
    w = (torch.randn(96, 512, generator=gen) * 0.05).to(DEV).requires_grad_(True)
    b = torch.randn(96, generator=gen).to(DEV).requires_grad_(True)
    v = torch.randn(4, 96, generator=gen).to(DEV)
    u = torch.randn(4, 512, generator=gen).to(DEV)

    def run(xx, ww, bb, lin):
        y = lin(xx, ww, bb)
        gx, = torch.autograd.grad((y * v.to(y.dtype)).sum() + (y ** 2).sum(), xx, create_graph=True)
        gw, gb, gx2 = torch.autograd.grad((gx * u.to(y.dtype)).sum(), [ww, bb, xx], allow_unused=True)
        return y, gx, gw, gx2

    got = run(x, w, b, lambda xx, ww, bb: equal_linear(xx, ww, bb, 0.3, 0.01) + scaled_mm(xx, ww.t(), 0.2))
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    ref = run(xd, wd, bd, lambda xx, ww, bb: 0.01 * bb + 0.3 * (xx @ ww.t()) + 0.2 * (xx @ ww.t()))
    for g_, r_ in zip(got, ref):
        assert rel_err(g_, r_) < 5e-6


@pytest.mark.parametrize('mode', ['f32', 'bf16x3'])
def test_resblock_blur_adjoint_fusion_gpu(mode):
    hip, _ = _be()
    prev, hip.conv_mode = hip.conv_mode, mode
    try:
        oc.check_resblock_blur_adjoint_fusion(DEV, size=256, batch=2, tol=1e-5 if mode == 'f32' else 2e-4)
    finally:
        hip.conv_mode = prev


@pytest.mark.parametrize('shape', [(2, 5, 129, 131), (1, 3, 64, 256), (3, 2, 70, 161)])
def test_upfirdn2d_mask_kernel(shape):
    """gc_upfirdn2d_mask_f32 == gc_upfirdn2d_f32 followed by gc_bias_act_bwd_f32, bit for bit (dense and row-pitched input)."""
    hip, _ = _be()
    gen = torch.Generator().manual_seed(sum(shape))
    b, c, h, w = shape
    k4 = torch.rand(4, 4, generator=gen).to(DEV)
    gy = torch.randn(b, c, h + 1, w + 1, generator=gen).to(DEV)
    ref_act = torch.randn(b, c, h, w, generator=gen).to(DEV)
    two = hip.bias_act_bwd(hip.upfirdn2d(gy, k4, 1, 1, 1, 1, h, w, False).contiguous(), ref_act, 0.2, 1.4)
    assert torch.equal(hip.upfirdn2d_mask(gy, k4, 1, 1, h, w, False, ref_act, 0.2, 1.4), two)
    assert torch.equal(hip.upfirdn2d_mask(_pitched(gy, (w + 1 + 31) // 32 * 32), k4, 1, 1, h, w, False, ref_act, 0.2, 1.4), two)


@pytest.mark.parametrize('shape,noise', [((2, 5, 128, 130), True), ((1, 3, 64, 256), False), ((3, 2, 70, 160), True), ((2, 4, 256, 256), True)])
def test_upfirdn2d_actbwd_kernel(shape, noise):
    """gc_upfirdn2d_actbwd_f32 (activation backward + bias / noise sums + Blur adjoint in one pass over the gradient) against
    gc_bias_act_bwd_reduce_f32 followed by gc_upfirdn2d_f32: the Blur adjoint bit for bit, the sums to fp32 summation order."""
    hip, _ = _be()
    gen = torch.Generator().manual_seed(sum(shape) + noise)
    b, c, h, w = shape
    k4 = torch.rand(4, 4, generator=gen).to(DEV)
    gy = torch.randn(b, c, h, w, generator=gen).to(DEV)
    y = torch.randn(b, c, h, w, generator=gen).to(DEV)
    nz = torch.randn(b, 1, h, w, generator=gen).to(DEV) if noise else None
    g_pre, psum, pdot, _ = hip.bias_act_bwd_reduce(gy, y, nz, 0.2, 1.4)
    ref = hip.upfirdn2d(g_pre, k4, 1, 1, 2, 2, h + 1, w + 1, False).contiguous()
    gx, ps, pd = hip.upfirdn2d_actbwd(gy, y, nz, k4, 2, 2, h + 1, w + 1, False, 0.2, 1.4)
    assert torch.equal(gx.contiguous(), ref)
    assert rel_err(ps.sum(2), psum.sum(2)) < 1e-5 and rel_err(ps.sum((0, 2)), g_pre.double().sum((0, 2, 3))) < 1e-5
    if noise:
        assert rel_err(pd.sum(), (g_pre.double() * nz.double()).sum()) < 1e-5
    else:
        assert pd is None


def test_upsampling_tail_backward_fused_vs_two_launches(bf16x3_mode):
    """upfirdn2d_bias_act's plain backward through the fused kernel and through the two-launch Functions: same input gradient (bit for bit),
    same bias / noise-strength gradients (summation order)."""
    import importlib
    upmod = importlib.import_module('gan_control_amd.models.op.upfirdn2d')
    from gan_control_amd.models.op import upfirdn2d_bias_act
    gen = torch.Generator().manual_seed(5)
    b, c, h = 2, 6, 129          # a (2H + 1)-wide transposed-convolution output
    k4 = (torch.rand(4, 4, generator=gen) + 0.1).to(DEV)
    res = []
    keep = upmod._FUSE_ACT_BWD
    try:
        for fused in (True, False):
            upmod._FUSE_ACT_BWD = fused
            gen2 = torch.Generator().manual_seed(6)
            x = torch.randn(b, c, h, h, generator=gen2).to(DEV).requires_grad_(True)
            bias = torch.randn(c, generator=gen2).to(DEV).requires_grad_(True)
            nw = torch.randn(1, generator=gen2).to(DEV).requires_grad_(True)
            nz = torch.randn(b, 1, h - 1, h - 1, generator=gen2).to(DEV)
            out = upfirdn2d_bias_act(x, k4, (1, 1), bias, nz, nw)
            go = torch.randn(out.shape, generator=gen2).to(DEV)
            res.append(torch.autograd.grad(out, [x, bias, nw], go))
    finally:
        upmod._FUSE_ACT_BWD = keep
    assert torch.equal(res[0][0], res[1][0])
    assert rel_err(res[0][1], res[1][1]) < 1e-5 and rel_err(res[0][2], res[1][2]) < 1e-5


@pytest.mark.parametrize('shape', [(2, 3, 32, 256, 260), (1, 3, 40, 512, 512), (3, 1, 16, 300, 301), (2, 2, 64, 256, 256)])
def test_pointwise_activation_backward_kernels(shape):
    """gc_pw_act_wgrad_f32 / gc_pw_act_dgrad_f32 (FromRGB's backward with the activation mask applied in the loads) against the three
    launches they replace: activation backward, pointwise weight gradient / input gradient, bias sum."""
    from gan_control_amd.models.op._backend import ConvGeom
    hip, emu = _be()
    b, k, n, h, w = shape
    gen = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(b, k, h, w, generator=gen).to(DEV)
    dy = torch.randn(b, n, h, w, generator=gen).to(DEV)
    y = torch.randn(b, n, h, w, generator=gen).to(DEV)
    w_adj = torch.randn(1, 1, n, k, generator=gen).to(DEV)
    assert hip.pw_act_supported(x, dy) == (b * h * w >= (1 << 18))          # the autograd layer only takes this path on bandwidth-sized planes
    g_pre = hip.bias_act_bwd(dy, y, 0.2, 1.4)
    dw, db = hip.pw_act_wgrad(x, dy, y, 0.2, 1.4)
    ref_dw = torch.einsum('bkp,bnp->kn', x.double().flatten(2), g_pre.double().flatten(2))
    assert dw.shape == (1, 1, k, n) and rel_err(dw[0, 0], ref_dw) < 1e-5
    assert rel_err(db, g_pre.double().sum((0, 2, 3))) < 1e-5
    gx = hip.pw_act_dgrad(dy, y, w_adj, 0.2, 1.4)
    geom = ConvGeom(1, 1, 1, 1, 0, 0, h, w)
    assert rel_err(gx, hip.conv2d(g_pre, w_adj, None, None, geom)) < 1e-6


@pytest.mark.parametrize('mode', ['f32', 'bf16x3'])
def test_from_rgb_backward_fused_vs_three_launches(mode):
    """D's first layer through _GConvAct.backward with and without the fused pointwise path: same gradients."""
    from gan_control_amd.models.op import conv2d_gradfix
    hip, _ = _be()
    prev, hip.conv_mode = hip.conv_mode, mode
    keep = conv2d_gradfix._FUSE_PW_ACT
    try:
        res = []
        for fused in (True, False):
            conv2d_gradfix._FUSE_PW_ACT = fused
            gen = torch.Generator().manual_seed(2)
            x = torch.randn(2, 3, 512, 512, generator=gen).to(DEV).requires_grad_(True)
            wt = (torch.randn(32, 3, 1, 1, generator=gen) * 0.5).to(DEV).requires_grad_(True)
            bias = torch.randn(32, generator=gen).to(DEV).requires_grad_(True)
            out = conv2d_gradfix.conv2d_bias_act(x, wt, bias, weight_scale=0.57)
            go = torch.randn(out.shape, generator=gen).to(DEV)
            res.append(torch.autograd.grad(out, [x, wt, bias], go))
        for a, c in zip(*res):
            assert rel_err(a, c) < 1e-5
    finally:
        conv2d_gradfix._FUSE_PW_ACT = keep
        hip.conv_mode = prev


SAMPLE_WGRAD_CASES = [
    # b, K, N, h, w, k, down, pad, pitched x
    (3, 64, 64, 48, 80, 3, 1, 1, False),       # 64k x 64n tiles, stride 1
    (2, 32, 48, 70, 66, 3, 1, 1, False),       # 32 x 32 tiles of six rows, ragged channel tile
    (4, 64, 96, 40, 40, 1, 1, 0, False),       # 1 x 1 on the matrix kernels
    (2, 64, 128, 65, 65, 3, 2, 0, False),      # stride 2, 64k x 64n, one output row per tile
    (3, 32, 64, 129, 129, 3, 2, 0, True),      # stride 2, 32k x 64n; x row-pitched as the transposed convolution leaves it
    (5, 64, 64, 6, 6, 3, 1, 1, False),         # fewer tiles per sample than the splits wanted
    (2, 128, 3, 128, 128, 1, 1, 0, False),     # ToRGB class: thin 1 x 1 on the vector ALUs
    (2, 3, 64, 128, 128, 1, 1, 0, False),      # FromRGB class
    (2, 160, 160, 64, 64, 3, 1, 1, False),     # 57 / 29 splits of a < 256 K-element tensor: the reduce passes with four lane groups per element group
]


@pytest.mark.parametrize('mode', ['bf16x3', 'bf16', 'f32'])
@pytest.mark.parametrize('case', SAMPLE_WGRAD_CASES)
def test_wgrad_samples_kernels(case, mode):
    """gc_conv2d_wgrad_samples_*: dw is the plain weight gradient, dw_samples[b] the weight gradient of sample b alone, and
    gc_wgrad_samples_contract_f32 turns the shares into the gradients of the two scales (checked against fp64 and against the
    plane products the route replaces)."""
    from gan_control_amd.models.op._backend import ConvGeom
    hip, emu = _be()
    b, K, N, h, w, k, down, pad, pitched = case
    gen = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    oh, ow = (h + 2 * pad - k) // down + 1, (w + 2 * pad - k) // down + 1
    geom = ConvGeom(k, k, 1, down, pad, pad, oh, ow)
    x = torch.randn(b, K, h, w, generator=gen).to(DEV)
    dy = torch.randn(b, N, oh, ow, generator=gen).to(DEV)
    si = (torch.randn(b, K, generator=gen) + 1.5).to(DEV)
    so = (torch.rand(b, N, generator=gen) + 0.5).to(DEV)
    wt = torch.randn(k, k, K, N, generator=gen).to(DEV)
    if pitched:
        pitch = (w + 31) // 32 * 32
        buf = torch.zeros(b, K, h, pitch, device=DEV)
        buf[..., :w] = x
        x = buf[..., :w]
    prev, hip.conv_mode = hip.conv_mode, mode
    try:
        thin = k == 1 and min(K, N) <= 4
        if mode == 'f32' and not thin:
            assert hip.conv2d_wgrad_samples_bytes(x, dy, geom) == 0          # fp32 arithmetic: the thin 1 x 1 shapes only
            with pytest.raises(ValueError):
                hip.conv2d_wgrad_samples(x, dy, si, so, geom)
            return
        assert hip.conv2d_wgrad_samples_bytes(x, dy, geom) > 0
        tol = 2e-2 if (mode == 'bf16' and not thin) else 5e-5
        for scales in ((si, so), (None, None), (si, None)):
            dw, dws = hip.conv2d_wgrad_samples(x, dy, scales[0], scales[1], geom)
            plain = hip.conv2d_wgrad(x, dy, scales[0], scales[1], geom)
            # same products, the splits grouped differently -- except where the batch gradient runs in exact fp32 (3 x 3 onto planes <= 8 x 8,
            # wgrad_f32_small_kernel) while the per-sample kernels keep the mode's arithmetic: then they differ by that arithmetic's error
            same = 1e-5 if not (k == 3 and max(oh, ow) <= 8) else tol
            assert rel_err(dw, plain) < same
            one = torch.stack([hip.conv2d_wgrad(x[i:i + 1].contiguous(), dy[i:i + 1], None if scales[0] is None else scales[0][i:i + 1],
                                                None if scales[1] is None else scales[1][i:i + 1], geom) for i in range(b)])
            assert rel_err(dws, one) < same
            assert rel_err(dws.sum(0), dw) < 1e-6
            ref = emu.conv2d_wgrad(x.double().cpu(), dy.double().cpu(), *[None if t is None else t.double().cpu() for t in scales], geom)
            assert rel_err(dw, ref) < tol
            d2, s2 = hip.conv2d_wgrad_samples(x, dy, scales[0], scales[1], geom)
            assert torch.equal(d2, dw) and torch.equal(s2, dws), 'run-to-run deterministic'
        dw, dws = hip.conv2d_wgrad_samples(x, dy, si, so, geom)
        g_a, g_c = hip.wgrad_samples_contract(dws, wt, si, so)
        prod = dws.double() * wt.double()
        assert rel_err(g_a, prod.sum((1, 2, 4)) / si.double()) < 1e-5
        assert rel_err(g_c, prod.sum((1, 2, 3)) / so.double()) < 1e-5
        only_a, none = hip.wgrad_samples_contract(dws, wt, si, None, True, False)
        assert none is None and torch.equal(only_a, g_a)
        zero = si.clone()
        zero[0, 1] = 0.0                                   # a scale of exactly zero divides by one (gc_rows_sum_div_f32's rule)
        g0, _ = hip.wgrad_samples_contract(dws, wt, zero, so, True, False)
        assert torch.equal(g0[0, 1], (g_a * si)[0, 1]) or rel_err(g0[0, 1], (g_a * si)[0, 1]) < 1e-6
        # ... and they are what the plane products give: d/dsi = sum_p x * (input gradient before its si factor), d/dso = sum_p dy * (y / so)
        if down == 1:
            y = hip.conv2d(x.contiguous(), wt, si, None, geom)
            assert rel_err(g_c, (dy.double() * y.double()).sum((2, 3))) < (5e-2 if mode == 'bf16' else 2e-4)
    finally:
        hip.conv_mode = prev


@pytest.mark.parametrize('mode', ['bf16x3', 'f32'])
def test_scale_grads_from_sample_wgrad(mode):
    """Generator backward and the path-length step with the modulation / demodulation gradients taken from the per-sample weight
    gradients against the plane-product route (f32 arithmetic: the ToRGB layers only have the form)."""
    hip, _ = _be()
    prev, hip.conv_mode = hip.conv_mode, mode
    try:
        kinds = oc.check_scale_grads_from_sample_wgrad(DEV, size=128, batch=3, tol=2e-4, pl_tol=2e-4, thin_only=mode == 'f32')
        assert (1, 1, 1) in kinds
    finally:
        hip.conv_mode = prev


@pytest.mark.parametrize('mode', ['bf16x3', 'f32'])
def test_torgb_fork(mode):
    """fp32 storage: the fork moves one addition from an ATen pass into a kernel epilogue and, where the per-sample route is not taken,
    subtracts it again for the modulation gradient -- rounding-level differences, which the path-length step's cancellation amplifies
    (5e-3: the size of either variant's own distance from fp64 there, tools/samples_route_probe.py)."""
    hip, _ = _be()
    prev, hip.conv_mode = hip.conv_mode, mode
    try:
        oc.check_torgb_fork(DEV, size=128, batch=3, tol=2e-4, pl_tol=5e-3)      # 2.4e-5 measured on the 4 x 4 ToRGB (gx - gfork cancels there)
    finally:
        hip.conv_mode = prev
