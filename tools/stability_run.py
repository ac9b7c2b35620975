#!/usr/bin/env python3
"""N training iterations at FFHQ-1024 (4 images, split-bf16 convolutions, every fusion at its default), losses every 16 iterations; every
statistic and every parameter must stay finite.  Usage: python tools/stability_run.py [iterations] [size] [batch]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from gan_control_amd.models.op import _backend
from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
size, batch = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1024, 4)
_backend.get().conv_mode = os.environ.get('GANCONTROL_CONV_PRECISION', 'bf16x3')
tr = GeneratorTrainer(default_config(size, batch), device='cuda', seed=0)
real = tr.synthetic_batch()
t0 = time.perf_counter()
for i in range(n):
    tr.train_iteration(i, real)
    if i % 16 == 15 or i == n - 1:
        st = {k: float(v) for k, v in tr.stats.items() if torch.is_tensor(v) and v.numel() == 1 or isinstance(v, (int, float))}
        assert all(v == v and abs(v) != float('inf') for v in st.values()), (i, st)
        print(i + 1, {k: round(v, 4) for k, v in st.items()}, flush=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
for name, net in (('G', tr.generator), ('D', tr.discriminator), ('G_ema', tr.g_ema)):
    bad = [k for k, v in net.state_dict().items() if v.dtype.is_floating_point and not torch.isfinite(v).all()]
    assert not bad, (name, bad[:4])
print(f'{n} iterations, {dt / n * 1e3:.1f} ms per iteration incl. the statistics read-back every 16; all statistics and parameters finite')
