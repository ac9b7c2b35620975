#!/usr/bin/env python3
"""The three weight re-layouts of a 3x3 convolution (parameter -> kernel layout, kernel layout -> input-gradient weights, weight gradient ->
parameter layout) and the hi / lo pack at the FFHQ-1024 channel counts: microseconds per pass with HIP events (dev tool)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from gan_control_amd.models.op import _backend
be = _backend.get()


def timeit(fn, reps=50):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for k, n in [(512, 512), (256, 512), (128, 256), (64, 128)]:
    taps = 9
    w = torch.randn(n, k, 3, 3, device='cuda')
    wt = torch.randn(3, 3, k, n, device='cuda')
    t1 = timeit(lambda: be.weight_layout(w, taps, k, n, (1, taps, k * taps), (3, 3, k, n), (k * n, n, 1), False, 0.5))
    t2 = timeit(lambda: be.weight_layout(wt, taps, k, n, (k * n, n, 1), (3, 3, n, k), (n * k, 1, k), True, 1.0))
    t3 = timeit(lambda: be.weight_layout(wt, taps, k, n, (k * n, n, 1), (n, k, 3, 3), (1, taps, k * taps), False, 0.5))
    print(f'{k:4d} -> {n:4d}: param->kernel {t1:6.1f} us   kernel->adjoint {t2:6.1f} us   grad->param {t3:6.1f} us', flush=True)
