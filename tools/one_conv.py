#!/usr/bin/env python3
"""Launch one convolution shape a few times (target for rocprofv3 --pmc passes).  usage: one_conv.py B K N res [k] [down]; ONE_CONV_WGRAD=1: its weight gradient"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from gan_control_amd.models.op import _backend
from gan_control_amd.models.op._backend import ConvGeom
be = _backend.get()
be.conv_mode = os.environ.get('GANCONTROL_CONV_PRECISION', 'bf16x3')
B, K, N, res = [int(v) for v in sys.argv[1:5]]
k = int(sys.argv[5]) if len(sys.argv) > 5 else 3
down = int(sys.argv[6]) if len(sys.argv) > 6 else 1
pad = k // 2 if down == 1 else 0
oh = (res + 2 * pad - k) // down + 1
g = ConvGeom(k, k, 1, down, pad, pad, oh, oh)
x = torch.randn(B, K, res, res, device='cuda'); w = torch.randn(k, k, K, N, device='cuda')
if os.environ.get('ONE_CONV_WGRAD'):       # the weight gradient of the same layer instead
    dy = torch.randn(B, N, oh, oh, device='cuda')
    for _ in range(3):
        y = be.conv2d_wgrad(x, dy, None, None, g)
else:
    for _ in range(3):
        y = be.conv2d(x, w, None, None, g)
torch.cuda.synchronize()
print('done')
