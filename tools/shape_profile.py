#!/usr/bin/env python3
"""Per-shape times of every convolution / weight-gradient launch over 16 training iterations (one full lazy-regulariser period)."""
import collections, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from gan_control_amd.models.op import _backend
from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config
from gan_control_amd.utils.profiling import conv_flops, conv_variant

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
be = _backend.get()
be.conv_mode = 'bf16x3'
tr = GeneratorTrainer(default_config(size, batch), device='cuda', seed=0)
real = tr.synthetic_batch()
for it in range(4):
    tr.train_iteration(it, real)
rec = collections.defaultdict(list)
orig_conv, orig_wg = be.conv2d, be.conv2d_wgrad


def wrap(kind, fn):
    def inner(x, second, in_scale, out_scale, geom, *rest, **kw):
        n_out = second.shape[3] if kind == 'conv' else second.shape[1]
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn(x, second, in_scale, out_scale, geom, *rest, **kw)
        e.record()
        key = (kind, x.shape[0], x.shape[1], n_out, x.shape[2], geom.kh, geom.up, geom.down,
               conv_variant(geom, n_out, x.shape[0], x.shape[1], 'bf16x3') if kind == 'conv' else '')
        rec[key].append((s, e, conv_flops(x.shape[0], x.shape[1], n_out, x.shape[2], x.shape[3], geom)))
        return out
    return inner


be.conv2d, be.conv2d_wgrad = wrap('conv', orig_conv), wrap('wgrad', orig_wg)
iters = 16
for it in range(16, 16 + iters):
    tr.train_iteration(it, real)
torch.cuda.synchronize()
rows = []
for key, evs in rec.items():
    ms = sum(s.elapsed_time(e) for s, e, _ in evs)
    rows.append((ms / iters, len(evs) / iters, 1e3 * ms / len(evs), evs[0][2] / (1e9 * ms / len(evs)), key))
tot = sum(r[0] for r in rows)
print(f'total {tot:.2f} ms/iter in conv + wgrad launches')
for ms, n, us, tf, key in sorted(rows, key=lambda r: -r[0]):
    kind, b, k, n_out, h, kh, up, down, var = key
    print(f'{ms:6.2f} ms/it {n:5.2f}/it {us:8.1f} us {tf:7.1f} TF  {kind:5s} B{b} {k:4d}->{n_out:4d} @{h:4d} k{kh} up{up} down{down} {var}')
