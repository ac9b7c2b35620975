#!/usr/bin/env python3
"""Random convolution shapes through every dispatcher branch (forward, with and without per-sample scales and a fused epilogue, and the
weight gradient) against fp64 ATen on the GPU (dev tool).  usage: fuzz_conv.py [cases] [seed] [kinds,comma,separated]; prints the kernel variants that were hit
and every case above the tolerance of its mode (f32 5e-6, bf16x3 5e-5)."""
import collections, os, random, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
import torch.nn.functional as F
from gan_control_amd.models.op import _backend
from gan_control_amd.models.op._backend import ConvGeom
from gan_control_amd.utils.profiling import conv_variant

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
KINDS = sys.argv[3].split(',') if len(sys.argv) > 3 else ['s1', 's1', 's1', 's2', 'up', 'small', 'small', 'pw', 'ws', 'ws', 'bigpw']
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
be = _backend.get()
dev = 'cuda'


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))


def reference(x, w, si, so, k, up, down, pad):
    xs = x.double() * (si.double()[:, :, None, None] if si is not None else 1.0)
    wk = w.double().permute(3, 2, 0, 1)                     # [N, K, kh, kw], correlation order
    if up == 2:
        xu = torch.zeros(x.shape[0], x.shape[1], 2 * x.shape[2] - 1, 2 * x.shape[3] - 1, dtype=torch.float64, device=x.device)
        xu[:, :, ::2, ::2] = xs
        y = F.conv2d(xu, wk, padding=pad)
    else:
        y = F.conv2d(xs, wk, stride=down, padding=pad)
    return y * (so.double()[:, :, None, None] if so is not None else 1.0)


hit = collections.Counter()
bad = 0
for c in range(cases):
    kind = rng.choice(KINDS)
    k = rng.choice([1, 3, 3, 3])
    up = down = 1
    if kind == 'small':
        h, w_ = rng.randint(1, 8), rng.randint(1, 8)
        b = rng.randint(1, 8)
        K, N = rng.choice([64, 72, 96, 128, 513, 512]), rng.choice([64, 70, 128, 513, 512])
        if rng.random() < 0.4:
            down = 2; h, w_ = 2 * h + k - 2 + rng.randint(0, 1), 2 * w_ + k - 2 + rng.randint(0, 1)
    elif kind == 'pw':
        k = 1; h, w_ = rng.randint(8, 300), rng.randint(8, 300); b = rng.randint(1, 4)
        K, N = rng.choice([(3, 32), (32, 3), (64, 3), (3, 64), (1, 16), (40, 2)])
    elif kind == 'bigpw':
        k = 1; h, w_ = rng.choice([512, 515, 600]), rng.choice([512, 511, 700]); b = rng.randint(1, 2)
        K, N = rng.choice([(3, 32), (32, 3), (20, 2), (3, 33)])
    elif kind == 's2ws':       # round 6: shapes for conv_s2ws_bf16x3_kernel (3x3, stride 2, pad 0, K % 16 == 0, N % 64 == 0, >= 192 tiles x samples x blocks)
        k = 3; down = 2
        h, w_ = rng.randint(130, 420), rng.randint(130, 420)
        b = rng.randint(1, 6)
        K, N = rng.choice([32, 48, 64, 80, 128, 256]), rng.choice([64, 128, 192, 256, 320])
    elif kind == 'ws':
        h, w_ = rng.choice([64, 65, 100, 128, 130, 150, 200, 257]), rng.choice([64, 96, 129, 132, 170, 190, 259, 262])
        b = rng.randint(2, 4)
        K, N = rng.choice([32, 48, 64, 80, 128, 256]), rng.choice([32, 64, 96, 128, 192])
        k = rng.choice([1, 3, 3])
    else:
        h, w_ = rng.choice([9, 16, 17, 31, 32, 33, 40, 64, 65, 100, 129, 130, 200, 257]), rng.choice([9, 16, 17, 32, 33, 48, 64, 65, 96, 129, 190, 259])
        b = rng.randint(1, 4)
        K, N = rng.choice([16, 24, 32, 48, 64, 80, 128, 256]), rng.choice([16, 32, 40, 64, 96, 128, 192])
        if kind == 's2': down = 2
        if kind == 'up': up = 2; k = 3; h, w_ = min(h, 70), min(w_, 70)
    pad = (2 if up == 2 else (k // 2 if down == 1 else 0))
    if b * K * h * w_ > 2 ** 25 or b * N * h * w_ * (4 if up == 2 else 1) > 2 ** 25:
        continue
    oh = (2 * h + 1 if up == 2 else (h + 2 * pad - k) // down + 1)
    ow = (2 * w_ + 1 if up == 2 else (w_ + 2 * pad - k) // down + 1)
    if oh < 1 or ow < 1:
        continue
    g = ConvGeom(k, k, up, down, pad, pad, oh, ow)
    gen = torch.Generator(device=dev).manual_seed(c)
    x = torch.randn(b, K, h, w_, device=dev, generator=gen); w = torch.randn(k, k, K, N, device=dev, generator=gen)
    si = torch.randn(b, K, device=dev, generator=gen); so = torch.rand(b, N, device=dev, generator=gen) + 0.5
    bias = torch.randn(N, device=dev, generator=gen)
    for mode, tol in (('f32', 5e-6), ('bf16x3', 5e-5)):
        be.conv_mode = mode
        hit[conv_variant(g, N, b, K, mode, (h, w_))] += 1
        for scales in ((None, None), (si, so)):
            ref = reference(x, w, scales[0], scales[1], k, up, down, pad if up == 1 else 2)
            out = be.conv2d(x, w, scales[0], scales[1], g).contiguous()
            e = rel(out, ref)
            if not e < tol:
                bad += 1; print('FWD', mode, (b, K, N, h, w_, k, up, down, pad), 'scales' if scales[0] is not None else 'plain', e, flush=True)
            fused = be.conv2d(x, w, scales[0], scales[1], g, epilogue=(bias, None, None, 0.2, 2 ** 0.5, True)).contiguous()
            eref = F.leaky_relu(ref + bias.double()[None, :, None, None], 0.2) * 2 ** 0.5
            e = rel(fused, eref)
            if not e < tol:
                bad += 1; print('EPI', mode, (b, K, N, h, w_, k, up, down, pad), 'scales' if scales[0] is not None else 'plain', e, flush=True)
        if up == 1:
            dy = torch.randn(b, N, oh, ow, device=dev, generator=gen)
            xs = (x.double() * si.double()[:, :, None, None]).requires_grad_(False)
            wref = torch.zeros(N, K, k, k, dtype=torch.float64, device=dev, requires_grad=True)
            (F.conv2d(xs, wref, stride=down, padding=pad) * (dy.double() * so.double()[:, :, None, None])).sum().backward()
            ref = wref.grad.permute(2, 3, 1, 0)
            out = be.conv2d_wgrad(x, dy, si, so, g)
            e = rel(out, ref)
            if not e < tol:
                bad += 1; print('WGRAD', mode, (b, K, N, h, w_, k, up, down, pad), e, flush=True)
print('variants hit:')
for n, cnt in sorted(hit.items()):
    print('  %4d  %s' % (cnt, n))
print('cases above tolerance:', bad)
