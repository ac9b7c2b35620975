#!/usr/bin/env python3
"""Dev tool (GPU): host (CPU) time per operator / autograd node over a few plain iterations (torch.profiler, CPU activity only) -- both the main
thread and the autograd engine's worker thread, which cProfile (tools/host_profile.py) cannot see."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from torch.profiler import profile, ProfilerActivity
from gan_control_amd.models.op import _backend
from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config
_backend.get().conv_mode = 'bf16x3'
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
tr = GeneratorTrainer(default_config(1024, batch), device='cuda', seed=0)
real = tr.synthetic_batch()
for i in range(1, 4):
    tr.train_iteration(i, real)
torch.cuda.synchronize()
its = [5, 6, 7, 9, 10, 11]
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for i in its:
        tr.train_iteration(i, real)
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.self_cpu_time_total)
tot = sum(e.self_cpu_time_total for e in rows)
print('total self CPU time per iteration: %.2f ms' % (tot / len(its) / 1e3))
for e in rows[:45]:
    print('%-70s calls/it %7.1f  self %8.3f ms/it  total %8.3f ms/it' % (e.key[:70], e.count / len(its), e.self_cpu_time_total / len(its) / 1e3, e.cpu_time_total / len(its) / 1e3))
