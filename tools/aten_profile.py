#!/usr/bin/env python3
"""Where does the non-HIP-library GPU time of one training iteration go?  (dev tool, GPU only)

Profiles one FFHQ-config iteration with torch.profiler and prints ATen ops grouped by (op, input shapes), sorted by
device time, so that the remaining PyTorch-side elementwise / copy / reduction passes can be fused away one by one.
"""
import argparse, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from torch.profiler import profile, ProfilerActivity
from gan_control_amd.models.op import _backend
from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=1024); ap.add_argument('--batch', type=int, default=4)
ap.add_argument('--precision', default='bf16x3'); ap.add_argument('--top', type=int, default=60)
ap.add_argument('--iter', type=int, default=16, help='iteration index to profile (16: with R1 + path-length, 17: plain D + G step)')
ap.add_argument('--all', action='store_true', help='include the HIP-library kernels and autograd Function rows')
a = ap.parse_args()
_backend.get().conv_mode = a.precision
cfg = default_config(a.size, a.batch)
tr = GeneratorTrainer(cfg, device='cuda', seed=0)
real = torch.randn(a.batch, 3, a.size, a.size, device='cuda').clamp(-1, 1)
for i in range(3):
    tr.train_iteration(i, real)          # includes the every-16 regularisers at i = 0
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.train_iteration(a.iter, real)     # 16: R1 and path-length passes included; 17: plain
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, 'self_device_time_total', None)
    if dt is None:
        dt = e.self_cuda_time_total
    if dt > 0 and (a.all or e.key.startswith('aten::')):
        rows.append((dt, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f'total device time {tot / 1e3:.1f} ms over {sum(r[1] for r in rows)} op calls')
for dt, n, k, sh in rows[:a.top]:
    print(f'{dt / 1e3:8.2f} ms {dt / tot * 100:5.1f}% {n:5d}x  {k[:40]:40s} {sh}')
