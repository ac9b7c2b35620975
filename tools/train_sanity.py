#!/usr/bin/env python3
"""Short real training run on synthetic data (dev tool): losses must stay finite and D must learn to separate."""
import argparse, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from gan_control_amd.models.op import _backend
from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=64); ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--iters', type=int, default=200); ap.add_argument('--precision', default='bf16x3')
ap.add_argument('--ada', action='store_true')
a = ap.parse_args()
_backend.get().conv_mode = a.precision
cfg = default_config(a.size, a.batch)
if a.ada:
    cfg['training_config']['augment'] = {'enabled': True, 'ada_target': 0.6, 'ada_length': 2000, 'p': 0}
tr = GeneratorTrainer(cfg, device='cuda', seed=0)
# "dataset": smooth low-frequency images so D has something to learn
gen = torch.Generator(device='cuda').manual_seed(1)
def batch():
    low = torch.randn(a.batch, 3, 4, 4, device='cuda', generator=gen)
    return torch.nn.functional.interpolate(low, size=a.size, mode='bilinear', align_corners=False).clamp(-1, 1)
for i in range(a.iters):
    tr.train_iteration(i, batch())
    if i % 25 == 0 or i == a.iters - 1:
        s = tr.reduced_stats()
        print(i, {k: round(v, 4) for k, v in s.items()}, flush=True)
        assert all(v == v and abs(v) < 1e6 for v in s.values()), 'non-finite statistic'
print('ok')
