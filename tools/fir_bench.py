#!/usr/bin/env python3
"""upfirdn2d shapes outside the 4x4 up=down=1 fast path (dev tool, GPU only)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from gan_control_amd.models.op import _backend
be = _backend.get()
k4 = torch.ones(4, 4, device='cuda') / 16


def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for B, c, res in [(8, 32, 1024), (4, 32, 1024), (8, 64, 512), (8, 128, 256), (8, 512, 64), (8, 512, 16)]:
    x = torch.randn(B, c, res, res, device='cuda')
    o = res // 2
    us = t(lambda: be.upfirdn2d(x, k4, 1, 2, 1, 1, o, o, True))
    by = 4.0 * (x.numel() + B * c * o * o)
    print(f'fir down2 {B}x{c}x{res}: {us:8.1f} us {by / us / 1e3:8.1f} GB/s')
    g = torch.randn(B, c, o, o, device='cuda')
    us = t(lambda: be.upfirdn2d(g, k4, 2, 1, 2, 2, res, res, False))
    print(f'fir up2 (adjoint) -> {res}: {us:8.1f} us {by / us / 1e3:8.1f} GB/s')
for B, c, res in [(4, 3, 512), (4, 3, 256)]:
    x = torch.randn(B, c, res, res, device='cuda')
    us = t(lambda: be.upfirdn2d(x, k4 * 4, 2, 1, 2, 2, 2 * res, 2 * res, True))
    by = 4.0 * (x.numel() * 5)
    print(f'rgb upsample {B}x{c}x{res}: {us:8.1f} us {by / us / 1e3:8.1f} GB/s')
