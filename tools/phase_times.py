#!/usr/bin/env python3
"""Wall time of the four phases of an iteration (D step, R1, G step, path length) with the GPU drained in between (dev tool)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from gan_control_amd.models.op import _backend
from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config
from gan_control_amd.trainers.utils import requires_grad, make_mini_batch_from_noise
_backend.get().conv_mode = os.environ.get('GANCONTROL_CONV_PRECISION', 'bf16x3')
size, batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 4
tr = GeneratorTrainer(default_config(size, batch), device='cuda', seed=0)
real = tr.synthetic_batch()
for i in range(2):
    tr.train_iteration(i * 16, real)
def timed(fn, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def dstep():
    requires_grad(tr.generator, False); requires_grad(tr.discriminator, True)
    tr.discriminator_step(make_mini_batch_from_noise(tr.sample_z(batch), batch, batch), [real])
def r1():
    requires_grad(tr.generator, False); requires_grad(tr.discriminator, True)
    tr.discriminator_regularize_step([real])
def gstep():
    requires_grad(tr.generator, True); requires_grad(tr.discriminator, False)
    tr.generator_step(make_mini_batch_from_noise(tr.sample_z(batch), batch, batch))
def pl():
    requires_grad(tr.generator, True); requires_grad(tr.discriminator, False)
    tr.generator_regularize_step()
t = {'d_step': timed(dstep), 'r1': timed(r1), 'g_step': timed(gstep), 'pl': timed(pl)}
print({k: round(v, 2) for k, v in t.items()}, 'ms; per iteration = d + g + r1/16 + pl/4 = %.2f ms' % (t['d_step'] + t['g_step'] + t['r1'] / 16 + t['pl'] / 4))
# CPU issue time (no synchronisation inside the timed region): a phase is launch-bound when this approaches its wall time
def issue(fn, n=5):
    out = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); out.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    return min(out)
print('CPU issue time (ms):', {k: round(issue(f), 2) for k, f in (('d_step', dstep), ('r1', r1), ('g_step', gstep), ('pl', pl))})
# device-busy time: the sum of the kernel durations of one repetition (torch.profiler); wall - busy = what the host leaves idle
def busy(fn):
    from torch.profiler import profile, ProfilerActivity
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn(); torch.cuda.synchronize()
    ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    return sum(e.device_time for e in ev) / 1e3, len(ev)
print('device busy (ms, launches):', {k: tuple(round(v, 2) for v in busy(f)) for k, f in (('d_step', dstep), ('r1', r1), ('g_step', gstep), ('pl', pl))})
