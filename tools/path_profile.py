#!/usr/bin/env python3
"""Device time of the path-length regularisation step by kernel / op (dev tool, GPU only)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from torch.profiler import profile, ProfilerActivity
from gan_control_amd.models.op import _backend
from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config
_backend.get().conv_mode = 'bf16x3'
tr = GeneratorTrainer(default_config(1024, 4), device='cuda', seed=0)
real = torch.randn(4, 3, 1024, 1024, device='cuda').clamp(-1, 1)
tr.train_iteration(0, real)
for _ in range(2):
    tr.generator_regularize_step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.generator_regularize_step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, 'self_device_time_total', None)
    if dt is None: dt = e.self_cuda_time_total
    if dt > 0 and e.key.startswith('aten::'): rows.append((dt, e.count, e.key + ' ' + str(e.input_shapes)[:90]))
import time
t0 = time.perf_counter()
for _ in range(3):
    tr.generator_regularize_step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 3 * 1e3
all_dev = sum((getattr(e, 'self_device_time_total', None) or 0) for e in prof.key_averages())
n_all = sum(e.count for e in prof.key_averages() if (getattr(e, 'self_device_time_total', None) or 0) > 0)
print(f'wall {wall:.1f} ms per step; device time of ALL kernels {all_dev / 1e3:.1f} ms over {n_all} launching ops (device-bound if the two agree)')
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f'total {tot/1e3:.1f} ms, {sum(r[1] for r in rows)} launches')
for dt, n, k in rows[:40]:
    print(f'{dt/1e3:8.2f} ms {dt/tot*100:5.1f}% {n:5d}x  {k[:110]}')
