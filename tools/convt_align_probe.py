#!/usr/bin/env python3
"""Dev probe (GPU): does the unaligned (2H+1)-wide output row cost the fused transposed convolution its store bandwidth?
The same launch with the output cropped to 2H columns (rows then start on 16-byte boundaries; the last column is simply not written)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from gan_control_amd.models.op import _backend
from gan_control_amd.models.op._backend import ConvGeom
be = _backend.get(); be.conv_mode = 'bf16x3'
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for B, ic, oc, res in [(4, 64, 32, 512), (4, 128, 64, 256), (4, 256, 128, 128)]:
    x = torch.randn(B, ic, res, res, device='cuda'); w = torch.randn(3, 3, ic, oc, device='cuda')
    for ow in (2 * res + 1, 2 * res, 2 * res - 31):
        g = ConvGeom(3, 3, 2, 1, 2, 2, 2 * res + 1, ow)
        print(f'convT {ic}->{oc} @{res}: output {2 * res + 1} x {ow}: {t(lambda: be.conv2d(x, w, None, None, g)):7.1f} us')
