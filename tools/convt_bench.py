#!/usr/bin/env python3
"""Up-sampling (transposed) convolution with and without the modulation scales (dev tool, GPU only)."""
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


for B, K, N, res in [(4, 512, 256, 64), (4, 256, 128, 128), (4, 128, 64, 256), (4, 64, 32, 512)]:
    oh = 2 * res + 1
    g = ConvGeom(3, 3, 2, 1, 2, 2, oh, oh)
    x = torch.randn(B, K, res, res, device='cuda'); w = torch.randn(3, 3, K, N, device='cuda')
    si, so = torch.randn(B, K, device='cuda'), torch.rand(B, N, device='cuda') + 0.5
    print(f'convT {K}->{N} @{res}: plain {t(lambda: be.conv2d(x, w, None, None, g)):7.1f} us   modulated {t(lambda: be.conv2d(x, w, si, so, g)):7.1f} us')
for B, K, N, res in [(4, 512, 512, 64), (4, 128, 128, 256), (4, 32, 32, 1024)]:
    g = ConvGeom(3, 3, 1, 1, 1, 1, res, res)
    x = torch.randn(B, K, res, res, device='cuda'); w = torch.randn(3, 3, K, N, device='cuda')
    si, so = torch.randn(B, K, device='cuda'), torch.rand(B, N, device='cuda') + 0.5
    print(f'conv  {K}->{N} @{res}: plain {t(lambda: be.conv2d(x, w, None, None, g)):7.1f} us   modulated {t(lambda: be.conv2d(x, w, si, so, g)):7.1f} us')
