#!/usr/bin/env python3
"""Host (CPU) time per operator / autograd node of one phase, from torch.profiler's CPU activities (dev tool, GPU only): which ATen calls and
which custom backward nodes the host spends its issue time in.  usage: host_ops_profile.py [d_step|r1|g_step|pl] [top]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402
from gan_control_amd.models.op import _backend  # noqa: E402
from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config  # noqa: E402
from gan_control_amd.trainers.utils import requires_grad, make_mini_batch_from_noise  # noqa: E402

phase = sys.argv[1] if len(sys.argv) > 1 else 'pl'
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
_backend.get().conv_mode = 'bf16x3'
size, batch = 1024, 4
tr = GeneratorTrainer(default_config(size, batch), device='cuda', seed=0)
real = tr.synthetic_batch()
for i in range(2):
    tr.train_iteration(i * 16, real)


def run():
    if phase == 'd_step':
        requires_grad(tr.generator, False); requires_grad(tr.discriminator, True)
        tr.discriminator_step(make_mini_batch_from_noise(tr.sample_z(batch), batch, batch), [real])
    elif phase == 'r1':
        requires_grad(tr.generator, False); requires_grad(tr.discriminator, True)
        tr.discriminator_regularize_step([real])
    elif phase == 'g_step':
        requires_grad(tr.generator, True); requires_grad(tr.discriminator, False)
        tr.generator_step(make_mini_batch_from_noise(tr.sample_z(batch), batch, batch))
    else:
        requires_grad(tr.generator, True); requires_grad(tr.discriminator, False)
        tr.generator_regularize_step()


run(); torch.cuda.synchronize()
n = 3
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(n):
        run()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.self_cpu_time_total)
tot = sum(e.self_cpu_time_total for e in rows)
print('phase %s: %.2f ms of host time per repetition in %d operator calls' % (phase, tot / n / 1e3, sum(e.count for e in rows) / n))
for e in rows[:top]:
    print('%8.2f ms %5.1f%% %6.0fx  %6.1f us/call  %s' % (e.self_cpu_time_total / n / 1e3, 100 * e.self_cpu_time_total / tot, e.count / n, e.self_cpu_time_total / e.count, e.key[:70]))
