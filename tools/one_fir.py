#!/usr/bin/env python3
"""Launch the 4 x 4 FIR tile kernel a few times on one plane shape (target for rocprofv3 --pmc passes).
usage: one_fir.py B C res [variant]; variant: plain (Blur, dense in, pitched out) | mask (Blur adjoint x activation mask) | act (Blur + noise + bias + leaky-ReLU)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from gan_control_amd.models.op import _backend
be = _backend.get()
be.conv_mode = 'bf16x3'
B, C, res = [int(v) for v in sys.argv[1:4]]
variant = sys.argv[4] if len(sys.argv) > 4 else 'plain'
k4 = (torch.ones(4, 4, device='cuda') / 16)
x = torch.randn(B, C, res, res, device='cuda')
for _ in range(3):
    if variant == 'plain':
        y = be.upfirdn2d(x, k4, 1, 1, 1, 1, res - 1, res - 1, True)
    elif variant == 'mask':
        a = torch.randn(B, C, res - 1, res - 1, device='cuda')
        y = be.upfirdn2d_mask(x, k4, 1, 1, res - 1, res - 1, False, a, 0.2, 1.414)
    else:
        bias = torch.randn(C, device='cuda'); nz = torch.randn(B, 1, res - 1, res - 1, device='cuda'); nw = torch.randn(1, device='cuda')
        y = be.upfirdn2d_act(x, k4, 1, 1, res - 1, res - 1, True, bias, nz, nw, 0.2, 1.414)
torch.cuda.synchronize()
print('done')
