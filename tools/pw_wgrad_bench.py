#!/usr/bin/env python3
"""ToRGB / FromRGB weight gradients (pointwise kernels): timing and error against fp64 (dev tool)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from gan_control_amd.models.op import _backend
from gan_control_amd.models.op._backend import ConvGeom
be = _backend.get(); be.conv_mode = 'bf16x3'
for B, K, N, H in [(4, 32, 3, 1024), (4, 64, 3, 512), (4, 128, 3, 256), (8, 3, 32, 1024), (2, 3, 64, 512), (2, 128, 3, 256), (4, 40, 3, 300),
                   (2, 3, 128, 256), (4, 256, 3, 128), (2, 256, 3, 128), (4, 512, 3, 64), (2, 512, 3, 64), (4, 512, 3, 32), (4, 512, 3, 16), (4, 512, 3, 8), (8, 3, 512, 32)]:
    g = ConvGeom(1, 1, 1, 1, 0, 0, H, H)
    x = torch.randn(B, K, H, H, device='cuda'); dy = torch.randn(B, N, H, H, device='cuda')
    si, so = torch.rand(B, K, device='cuda') + 0.5, torch.rand(B, N, device='cuda') + 0.5
    fn = lambda: be.conv2d_wgrad(x, dy, si, so, g)
    out = fn(); torch.cuda.synchronize()
    ref = torch.einsum('bkp,bnp->kn', (x * si[:, :, None, None]).double().flatten(2), (dy * so[:, :, None, None]).double().flatten(2))
    err = (out.double().reshape(K, N) - ref).abs().max().item() / ref.abs().max().item()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    print(f'B{B} {K}->{N} @{H}: {s.elapsed_time(e) * 100:8.1f} us   rel err {err:.2e}', flush=True)
