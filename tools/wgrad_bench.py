#!/usr/bin/env python3
"""Times the split-bf16 weight-gradient kernels at the FFHQ-1024 layer shapes (dev tool; GANCONTROL_HIP_LIB picks the library)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from gan_control_amd.models.op import _backend
from gan_control_amd.models.op._backend import ConvGeom
from gan_control_amd.utils.profiling import conv_flops
be = _backend.get()
be.conv_mode = 'bf16x3'
SHAPES = [(8, 32, 32, 1024, 1), (8, 64, 64, 512, 1), (8, 128, 128, 256, 1), (8, 512, 512, 64, 1),
          (8, 32, 64, 1025, 2), (8, 64, 128, 513, 2), (8, 128, 256, 257, 2), (8, 256, 512, 129, 2), (8, 512, 512, 65, 2)]
for B, K, N, H, down in SHAPES:
    oh = H if down == 1 else (H - 3) // 2 + 1
    g = ConvGeom(3, 3, 1, down, 1 if down == 1 else 0, 1 if down == 1 else 0, oh, oh)
    x = torch.randn(B, K, H, H, device='cuda'); dy = torch.randn(B, N, oh, oh, device='cuda')
    fn = lambda: be.conv2d_wgrad(x, dy, None, None, g)
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 100
    print(f'B{B} {K:4d}->{N:4d} @{H:4d} down{down}: {us:8.1f} us {conv_flops(B, K, N, H, H, g) / us / 1e6:7.1f} TF', flush=True)
    del x, dy
