#!/usr/bin/env python3
"""Per-layer micro-benchmark of the HIP primitives at the FFHQ-1024 layer shapes (dev tool, GPU only).

Times each distinct conv / wgrad / FIR / bias-act shape of G and D (batch 4) with HIP events and
prints achieved TFLOP/s or GB/s against the roofline that bounds it.
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch  # noqa: E402
from gan_control_amd.models.op import _backend  # noqa: E402
from gan_control_amd.models.op._backend import ConvGeom  # noqa: E402
from gan_control_amd.utils.profiling import conv_flops, conv_variant  # noqa: E402


def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--only', default='')
    ap.add_argument('--mode', default='f32')
    args = ap.parse_args()
    be = _backend.get()
    be.conv_mode = args.mode
    B, dev = args.batch, 'cuda'
    rows = []
    convs = []
    for ch, res in [(512, 4), (512, 8), (512, 16), (512, 32), (512, 64), (256, 128), (128, 256), (64, 512), (32, 1024)]:
        convs.append(('conv3x3 s1', ch, ch, res, 3, 1, 1, 1))
    for (ic, oc, res) in [(512, 512, 4), (512, 512, 8), (512, 512, 16), (512, 512, 32), (512, 256, 64), (256, 128, 128), (128, 64, 256), (64, 32, 512)]:
        convs.append(('convT3x3 up2', ic, oc, res, 3, 2, 1, 2))
    for (ic, oc, res) in [(32, 64, 1025), (64, 128, 513), (128, 256, 257), (256, 512, 129), (512, 512, 65), (512, 512, 33), (512, 512, 17), (512, 512, 9)]:
        convs.append(('conv3x3 s2', ic, oc, res, 3, 1, 2, 0))
    for (ic, oc, res) in [(32, 64, 1023), (128, 256, 255), (512, 512, 63)]:
        convs.append(('conv1x1 s2', ic, oc, res, 1, 1, 2, 0))
    # the 1x1 convolutions of D's skip paths (after the FIR down-sampling) and their input gradients: HBM-bound
    for (ic, oc, res) in [(32, 64, 512), (64, 128, 256), (128, 256, 128), (256, 512, 64), (64, 32, 512), (128, 64, 256), (256, 128, 128)]:
        convs.append(('conv1x1 s1', ic, oc, res, 1, 1, 1, 0))
    convs.append(('conv1x1 torgb', 32, 3, 1024, 1, 1, 1, 0))
    convs.append(('conv1x1 fromrgb', 3, 32, 1024, 1, 1, 1, 0))
    for name, ic, oc, res, k, up, down, pad in convs:
        if args.only and args.only not in f'{name} {ic}->{oc} @{res}':
            continue
        if up > 1:
            oh = (res - 1) * up + k - 2 * (k - 1 - pad)
        else:
            oh = (res + 2 * pad - k) // down + 1
        g = ConvGeom(k, k, up, down, pad, pad, oh, oh)
        x = torch.randn(B, ic, res, res, device=dev)
        w = torch.randn(k, k, ic, oc, device=dev)
        fl = conv_flops(B, ic, oc, res, res, g)
        t = timeit(lambda: be.conv2d(x, w, None, None, g), args.reps)
        rows.append((f'{name} {ic}->{oc} @{res}', conv_variant(g, oc, B, ic, args.mode, (res, res)), t * 1e6, fl / t / 1e12, 'TF/s', fl / t / 1e12 / 157.3))
        if up == 1:
            dy = torch.randn(B, oc, oh, oh, device=dev)
            t = timeit(lambda: be.conv2d_wgrad(x, dy, None, None, g), args.reps)
            rows.append((f'wgrad {name} {ic}->{oc} @{res}', 'wgrad', t * 1e6, fl / t / 1e12, 'TF/s', fl / t / 1e12 / 157.3))
        del x, w
    if not args.only or 'fir' in args.only:
        k4 = torch.ones(4, 4, device=dev) / 16
        for c, res in [(32, 1025), (64, 513), (128, 257), (512, 65)]:
            x = torch.randn(B, c, res, res, device=dev)
            t = timeit(lambda: be.upfirdn2d(x, k4, 1, 1, 1, 1, res - 1, res - 1, True), args.reps)
            by = 4.0 * (x.numel() + B * c * (res - 1) ** 2)
            rows.append((f'fir44 {c}x{res}', 'fir44_tile', t * 1e6, by / t / 1e9, 'GB/s', by / t / 1e9 / 8000))
        for c, res in [(32, 1024), (64, 512), (128, 256), (512, 64), (32, 1025), (64, 513)]:
            x = torch.randn(B, c, res, res, device=dev)
            b = torch.randn(c, device=dev)
            nz = torch.randn(B, 1, res, res, device=dev)
            nw = torch.randn(1, device=dev)
            t = timeit(lambda: be.bias_act(x, b, nz, nw, 0.2, 1.414), args.reps)
            by = 4.0 * (2 * x.numel() + nz.numel())
            rows.append((f'bias_act+noise {c}x{res}', 'bias_act', t * 1e6, by / t / 1e9, 'GB/s', by / t / 1e9 / 8000))
            y = be.bias_act(x, b, None, None, 0.2, 1.414)
            t = timeit(lambda: be.bias_act_bwd(x, y, 0.2, 1.414), args.reps)
            by = 4.0 * 3 * x.numel()
            rows.append((f'bias_act_bwd {c}x{res}', 'bias_act_bwd', t * 1e6, by / t / 1e9, 'GB/s', by / t / 1e9 / 8000))
            t = timeit(lambda: be.bias_act_bwd_reduce(x, y, None, 0.2, 1.414), args.reps)
            rows.append((f'bias_act_bwd_reduce {c}x{res}', 'bias_act_bwd_reduce', t * 1e6, by / t / 1e9, 'GB/s', by / t / 1e9 / 8000))
            t = timeit(lambda: be.bias_act_bwd_reduce(x, y, nz, 0.2, 1.414, self_dot=(b, nw)), args.reps)
            rows.append((f'bias_act_bwd_reduce+noise+self {c}x{res}', 'bias_act_bwd_reduce', t * 1e6, (by + 4.0 * nz.numel()) / t / 1e9, 'GB/s', (by + 4.0 * nz.numel()) / t / 1e9 / 8000))
            t = timeit(lambda: be.channel_sum(x), args.reps)
            rows.append((f'channel_sum {c}x{res}', 'channel_sum', t * 1e6, 4.0 * x.numel() / t / 1e9, 'GB/s', 4.0 * x.numel() / t / 1e9 / 8000))
    for r in rows:
        print('%-44s %-46s %10.1f us %9.2f %s  %5.1f%%' % (r[0], r[1], r[2], r[3], r[4], 100 * r[5]))


if __name__ == '__main__':
    main()
