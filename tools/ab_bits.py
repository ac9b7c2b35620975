#!/usr/bin/env python3
"""Dev tool: run a few conv shapes and dump checksums (compare two library builds bit for bit).  usage: ab_bits.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from gan_control_amd.models.op import _backend
from gan_control_amd.models.op._backend import ConvGeom
be = _backend.get(); be.conv_mode = 'bf16x3'
gen = torch.Generator().manual_seed(5)
for (B, K, N, res, k, up, down, pad) in [(2, 32, 32, 256, 3, 1, 1, 1), (2, 64, 32, 128, 3, 2, 1, 2), (2, 32, 64, 257, 3, 1, 2, 0), (2, 64, 64, 128, 3, 1, 1, 1), (2, 128, 64, 64, 3, 2, 1, 2)]:
    oh = (res - 1) * up + k - 2 * (k - 1 - pad) if up > 1 else (res + 2 * pad - k) // down + 1
    g = ConvGeom(k, k, up, down, pad, pad, oh, oh)
    x = torch.randn(B, K, res, res, generator=gen).cuda(); w = torch.randn(k, k, K, N, generator=gen).cuda()
    si = torch.randn(B, K, generator=gen).cuda(); so = (torch.rand(B, N, generator=gen) + 0.5).cuda()
    y = be.conv2d(x, w, si, so, g)
    ref = be  # checksum only
    print((B, K, N, res, k, up, down), float(y.double().sum()), float(y.double().abs().sum()), int(y.view(torch.int32).sum(dtype=torch.int64)))
