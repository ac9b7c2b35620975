import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from gan_control_amd.models.op import _backend
be = _backend.get()
k4 = torch.ones(4, 4, device='cuda') / 16
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for B, c, res in [(8, 32, 1024), (8, 64, 512), (8, 128, 256), (4, 32, 1024)]:
    x = torch.randn(B, c, res, res, device='cuda')
    us = t(lambda: be.upfirdn2d(x, k4, 1, 1, 2, 2, res + 1, res + 1, True))
    by = 4.0 * (x.numel() + B * c * (res + 1) ** 2)
    print(f'blur {B}x{c}x{res} -> {res+1}: {us:8.1f} us {by/us/1e3:8.1f} GB/s')
