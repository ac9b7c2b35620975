#!/usr/bin/env python3
"""Dev probe (GPU): what do the unaligned (H+1)-wide output rows cost the Blur of D's down-sampling layers (fir44_tile_kernel)?"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from gan_control_amd.models.op import _backend
be = _backend.get()
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
k4 = torch.ones(4, 4, device='cuda') / 16
for B, c, res in [(8, 32, 1024), (8, 64, 512), (8, 128, 256)]:
    x = torch.randn(B, c, res, res, device='cuda')
    for ow in (res + 1, res):
        us = t(lambda: be.upfirdn2d(x, k4, 1, 1, 2, 2, res + 1, ow, True))
        print(f'fir44 [{B},{c},{res},{res}] -> {res + 1} x {ow}: {us:7.1f} us  {4.0 * (x.numel() + B * c * (res + 1) * ow) / us / 1e3:7.1f} GB/s')
