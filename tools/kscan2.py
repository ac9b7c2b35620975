#!/usr/bin/env python3
"""Does a power-of-two plane stride hurt the convolution's loads / stores?  Same layer at neighbouring plane sizes."""
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


for B, K, N in [(8, 32, 32), (8, 64, 64)]:
    for res in (960, 992, 1008, 1024, 1040, 1056, 512, 520, 528):
        if K == 64 and res > 600 or K == 32 and res < 600: continue
        g = ConvGeom(3, 3, 1, 1, 1, 1, res, res)
        x = torch.randn(B, K, res, res, device='cuda'); w = torch.randn(3, 3, K, N, device='cuda')
        us = t(lambda: be.conv2d(x, w, None, None, g))
        print(f'B{B} {K:4d}->{N:4d} @{res}: {us:8.1f} us   {us / (res * res) * 1e3:7.4f} ns/px   in+out {4e-3 * B * (K + N) * res * res / us:7.1f} GB/s')
        del x
