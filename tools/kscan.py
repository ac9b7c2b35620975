#!/usr/bin/env python3
"""Convolution time vs number of input channels at a fixed plane size: separates the per-tile fixed cost from the per-chunk cost."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from gan_control_amd.models.op import _backend
from gan_control_amd.models.op._backend import ConvGeom
be = _backend.get()
be.conv_mode = 'bf16x3'


def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for B, N, res in [(8, 32, 1024), (8, 64, 512), (4, 128, 256)]:
    for K in (16, 32, 64, 128, 256):
        g = ConvGeom(3, 3, 1, 1, 1, 1, res, res)
        x = torch.randn(B, K, res, res, device='cuda'); w = torch.randn(3, 3, K, N, device='cuda')
        us = t(lambda: be.conv2d(x, w, None, None, g))
        fl = 2.0 * B * K * N * 9 * res * res
        print(f'B{B} {K:4d}->{N:4d} @{res}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s   in+out {4e-3 * B * (K + N) * res * res / us:7.1f} GB/s')
        del x
