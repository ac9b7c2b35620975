#!/usr/bin/env python3
"""Activation-backward + reductions and plane-dot at the top-level shapes (dev tool, GPU only)."""
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
for B, c, res in [(8, 32, 1024), (4, 32, 1024), (8, 64, 512), (4, 512, 64)]:
    dy = torch.randn(B, c, res, res, device='cuda'); y = torch.randn_like(dy); nz = torch.randn(B, 1, res, res, device='cuda')
    bias = torch.randn(c, device='cuda'); nw = torch.randn(1, device='cuda')
    by = 12.0 * dy.numel()
    a = t(lambda: be.bias_act_bwd_reduce(dy, y, None, 0.2, 1.414))
    b_ = t(lambda: be.bias_act_bwd_reduce(dy, y, nz, 0.2, 1.414, self_dot=(bias, nw)))
    d = t(lambda: be.plane_dot(dy, y))
    print(f'{B}x{c}x{res}: bwd_reduce {a:7.1f} us {by/a/1e3:7.0f} GB/s | +noise+self {b_:7.1f} us {by/b_/1e3:7.0f} GB/s | plane_dot {d:7.1f} us {8.0*dy.numel()/d/1e3:7.0f} GB/s')
