#!/usr/bin/env python3
"""HBM traffic per launch from rocprofv3 PMC passes (dev tool, run under `rocprofv3 --pmc FETCH_SIZE` and again
under `--pmc WRITE_SIZE`).  Launches, once each: a float4 device copy of known size (calibration), the dominant
conv kernel shape, the FIR kernel and the fused bias-act kernel at their FFHQ-1024 shapes."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch  # noqa: E402
from gan_control_amd.models.op import _backend  # noqa: E402
from gan_control_amd.models.op._backend import ConvGeom  # noqa: E402

be = _backend.get()
be.conv_mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'
dev = 'cuda'
a = torch.randn(128 * 1024 * 1024, device=dev)          # 512 MiB
for _ in range(2):
    b = a.clone()                                        # calibration: 512 MiB read + 512 MiB write
x = torch.randn(4, 128, 256, 256, device=dev); w = torch.randn(3, 3, 128, 128, device=dev)
g = ConvGeom(3, 3, 1, 1, 1, 1, 256, 256)
for _ in range(2):
    y = be.conv2d(x, w, None, None, g)
    dw = be.conv2d_wgrad(x, y, None, None, g)
k4 = torch.ones(4, 4, device=dev) / 16
xf = torch.randn(4, 32, 1025, 1025, device=dev)
for _ in range(2):
    yf = be.upfirdn2d(xf, k4, 1, 1, 1, 1, 1024, 1024, True)
xb = torch.randn(4, 32, 1024, 1024, device=dev); bias = torch.randn(32, device=dev)
for _ in range(2):
    yb = be.bias_act(xb, bias, None, None, 0.2, 1.414)
torch.cuda.synchronize()
print('done')
