#!/usr/bin/env python3
"""Convolution with the fused epilogue vs convolution + gc_bias_act_f32 at the discriminator's layer shapes (dev tool, GPU only)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from gan_control_amd.models.op import _backend
from gan_control_amd.models.op._backend import ConvGeom
be = _backend.get()
be.conv_mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'


def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for B, K, N, res, k, down, pad in [(8, 3, 32, 1024, 1, 1, 0), (8, 32, 32, 1024, 3, 1, 1), (8, 32, 64, 1025, 3, 2, 0), (8, 64, 64, 512, 3, 1, 1), (8, 64, 128, 513, 3, 2, 0),
                                   (8, 128, 128, 256, 3, 1, 1), (8, 256, 256, 128, 3, 1, 1), (8, 512, 512, 64, 3, 1, 1), (8, 512, 512, 32, 3, 1, 1)]:
    oh = (res + 2 * pad - k) // down + 1
    g = ConvGeom(k, k, 1, down, pad, pad, oh, oh)
    x = torch.randn(B, K, res, res, device='cuda'); w = torch.randn(k, k, K, N, device='cuda'); b = torch.randn(N, device='cuda')
    t_plain = t(lambda: be.conv2d(x, w, None, None, g))
    y = be.conv2d(x, w, None, None, g)
    t_act = t(lambda: be.bias_act(y, b, None, None, 0.2, 1.414))
    t_fused = t(lambda: be.conv2d(x, w, None, None, g, epilogue=(b, None, None, 0.2, 1.414, True)))
    print(f'{K:4d}->{N:4d} @{res:5d} k{k} s{down}: conv {t_plain:8.1f} us + act {t_act:7.1f} us = {t_plain + t_act:8.1f}   fused {t_fused:8.1f} us')
