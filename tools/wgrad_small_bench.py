#!/usr/bin/env python3
"""Weight gradients on the small planes (4^2 .. 33^2, 512 channels): timing and error against an fp64 reference (dev tool)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
import torch.nn.functional as F
from gan_control_amd.models.op import _backend
from gan_control_amd.models.op._backend import ConvGeom
be = _backend.get(); be.conv_mode = 'bf16x3'
for B, K, N, H, down in [(8, 512, 512, 16, 1), (4, 512, 512, 16, 1), (8, 512, 512, 8, 1), (8, 512, 512, 4, 1), (8, 512, 512, 33, 2), (4, 512, 512, 33, 2), (8, 512, 512, 17, 2), (8, 512, 512, 9, 2), (8, 512, 512, 32, 1)]:
    oh = H if down == 1 else (H - 3) // 2 + 1
    pad = 1 if down == 1 else 0
    g = ConvGeom(3, 3, 1, down, pad, pad, oh, oh)
    x = torch.randn(B, K, H, H, device='cuda'); dy = torch.randn(B, N, oh, oh, device='cuda')
    fn = lambda: be.conv2d_wgrad(x, dy, None, None, g)
    out = fn(); torch.cuda.synchronize()
    ref = torch.nn.grad.conv2d_weight(x.double(), (N, K, 3, 3), dy.double(), stride=down, padding=pad).permute(2, 3, 1, 0)
    err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    print(f'B{B} {K}->{N} @{H} down{down}: {s.elapsed_time(e) * 100:8.1f} us   rel err {err:.2e}', flush=True)
