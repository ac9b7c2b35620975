#!/usr/bin/env python3
"""Device time of the four phases of a training iteration (dev tool, GPU only): D step, R1 step, G step, path-length step."""
import argparse, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from gan_control_amd.models.op import _backend
from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=1024); ap.add_argument('--batch', type=int, default=4)
ap.add_argument('--precision', default='bf16x3'); ap.add_argument('--reps', type=int, default=5)
a = ap.parse_args()
_backend.get().conv_mode = a.precision
tr = GeneratorTrainer(default_config(a.size, a.batch), device='cuda', seed=0)
real = torch.randn(a.batch, 3, a.size, a.size, device='cuda').clamp(-1, 1)
for i in range(2):
    tr.train_iteration(i, real)


def timed(fn):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / a.reps


tc = tr.config['training_config']
z = lambda: tr.sample_z(a.batch)
rows = [('D step (G fwd no-grad, D fwd+bwd on fake|real, Adam)', timed(lambda: tr.discriminator_step([z()], [real])), 1.0),
        ('R1 step (D fwd + double backward on real, Adam)', timed(lambda: tr.discriminator_regularize_step([real])), 1.0 / tc['d_reg_every']),
        ('G step (G fwd, D fwd, bwd through both, Adam)', timed(lambda: tr.generator_step([z()])), 1.0),
        ('path-length step (G fwd + double backward, Adam)', timed(lambda: tr.generator_regularize_step()), 1.0 / tc['g_reg_every'])]
tot = sum(t * w for _, t, w in rows)
for name, t, w in rows:
    print(f'{name:58s} {t:8.2f} ms  x{w:6.4f} = {t * w:7.2f} ms/iter ({t * w / tot * 100:4.1f}%)')
print(f'{"sum":58s} {tot:8.2f} ms/iter  -> {a.batch / tot * 1e3:.1f} img/s')
