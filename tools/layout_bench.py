#!/usr/bin/env python3
"""gc_weight_layout_f32 at the model's weight shapes (dev tool, GPU only)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from gan_control_amd.models.op import _backend
be = _backend.get()


def t(fn, reps=50):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for n, k, kh in [(512, 512, 3), (256, 256, 3), (64, 64, 3), (32, 32, 3), (512, 512, 1), (3, 32, 1)]:
    taps = kh * kh
    w = torch.randn(n, k, kh, kh, device='cuda')
    wt = torch.randn(kh, kh, k, n, device='cuda')
    a = t(lambda: be.weight_layout(w, taps, k, n, (1, taps, k * taps), (kh, kh, k, n), (k * n, n, 1), False, 0.5))
    b = t(lambda: be.weight_layout(wt, taps, k, n, (k * n, n, 1), (kh, kh, n, k), (n * k, 1, k), True, 1.0))
    c = t(lambda: be.weight_layout(wt, taps, k, n, (k * n, n, 1), (n, k, kh, kh), (1, taps, k * taps), False, 0.5))
    by = 8.0 * n * k * taps
    print(f'{n}x{k}x{kh}x{kh}: prep {a:6.1f} us ({by / a / 1e3:6.0f} GB/s)  adjoint {b:6.1f} us  unprep {c:6.1f} us')
