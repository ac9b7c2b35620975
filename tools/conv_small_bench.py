#!/usr/bin/env python3
"""Forward / transposed / stride-2 convolutions on the small planes (4^2 .. 33^2, 512 channels): timing and error against fp64 (dev tool)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
import torch.nn.functional as F
from gan_control_amd.models.op import _backend
from gan_control_amd.models.op._backend import ConvGeom
be = _backend.get(); be.conv_mode = 'bf16x3'
for B, K, N, H, up, down in [(4, 512, 512, 16, 1, 1), (8, 512, 512, 16, 1, 1), (4, 512, 512, 8, 1, 1), (4, 512, 512, 4, 1, 1), (4, 512, 512, 33, 1, 2), (8, 512, 512, 33, 1, 2),
                             (4, 512, 512, 17, 1, 2), (4, 512, 512, 9, 1, 2), (4, 512, 512, 8, 2, 1), (4, 512, 512, 4, 2, 1), (4, 512, 512, 16, 2, 1)]:
    if up == 2:
        oh = 2 * H + 1; g = ConvGeom(3, 3, 2, 1, 2, 2, oh, oh)
    elif down == 2:
        oh = (H - 3) // 2 + 1; g = ConvGeom(3, 3, 1, 2, 0, 0, oh, oh)
    else:
        oh = H; g = ConvGeom(3, 3, 1, 1, 1, 1, oh, oh)
    x = torch.randn(B, K, H, H, device='cuda'); w = torch.randn(3, 3, K, N, device='cuda') / 68
    fn = lambda: be.conv2d(x, w, None, None, g)
    out = fn(); torch.cuda.synchronize()
    wk = w.permute(3, 2, 0, 1).double()
    if up == 2:
        ref = F.conv_transpose2d(x.double(), wk.flip(2, 3).transpose(0, 1).contiguous(), stride=2)
    else:
        ref = F.conv2d(x.double(), wk, stride=down, padding=1 if down == 1 else 0)
    err = (out.double() - ref).abs().max().item() / ref.abs().max().item() if tuple(ref.shape) == tuple(out.shape) else float('nan')
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    print(f'B{B} {K}->{N} @{H} up{up} down{down}: {s.elapsed_time(e) * 100:8.1f} us   rel err {err:.2e}', flush=True)
