#!/usr/bin/env python3
"""Dev tool (GPU): the decimating (down = 2) and interpolating (up = 2) 4 x 4 FIR kernels at the step's shapes: microseconds and GB/s of
algorithmic traffic (read + write)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from gan_control_amd.models.op import _backend
be = _backend.get(); be.conv_mode = 'bf16x3'
k4 = torch.ones(4, 4, device='cuda') / 16
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3
for B in (4, 8):
    for c, res in [(32, 1024), (64, 512), (128, 256), (256, 128), (512, 64)]:
        x = torch.randn(B, c, res, res, device='cuda')
        t = timeit(lambda: be.upfirdn2d(x, k4, 1, 2, 1, 1, res // 2, res // 2, True))
        by = 4.0 * (x.numel() + x.numel() / 4)
        print(f'down2 B{B} {c}x{res}: {t*1e6:7.1f} us {by/t/1e9:7.0f} GB/s')
        y = torch.randn(B, c, res // 2, res // 2, device='cuda')
        t = timeit(lambda: be.upfirdn2d(y, k4 * 4, 2, 1, 2, 2, res, res, False))
        print(f'up2   B{B} {c}x{res//2}->{res}: {t*1e6:7.1f} us {by/t/1e9:7.0f} GB/s')
    x = torch.randn(B, 3, 512, 512, device='cuda')
    t = timeit(lambda: be.upfirdn2d(x, k4 * 4, 2, 1, 2, 2, 1024, 1024, True))
    print(f'up2   B{B} 3x512->1024: {t*1e6:7.1f} us {4.0*(x.numel()*5)/t/1e9:7.0f} GB/s')
