#!/usr/bin/env python3
"""Which upfirdn2d calls of a training iteration take the generic kernel, and what do they cost?  (dev tool, GPU only)"""
import collections, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from gan_control_amd.models.op import _backend
from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config
be = _backend.get(); be.conv_mode = 'bf16x3'
tr = GeneratorTrainer(default_config(1024, 4), device='cuda', seed=0)
real = tr.synthetic_batch()
for it in range(4): tr.train_iteration(it, real)
rec = collections.defaultdict(list)
orig = be.upfirdn2d
def wrapped(x, taps, up, down, pad_x0, pad_y0, out_h, out_w, flip):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); y = orig(x, taps, up, down, pad_x0, pad_y0, out_h, out_w, flip); e.record()
    fast = tuple(taps.shape) == (4, 4) and ((up == 1 and down == 1 and out_w >= 64 and out_h >= 16) or ((up, down) in ((1, 2), (2, 1)) and out_w >= 32 and out_h >= 8))
    rec[(fast, tuple(x.shape), tuple(taps.shape), up, down, pad_x0, pad_y0, out_h, out_w, int(flip))].append((s, e, 4.0 * (x.numel() + y.numel())))
    return y
be.upfirdn2d = wrapped
for it in range(16, 32): tr.train_iteration(it, real)
torch.cuda.synchronize()
rows = []
for k, evs in rec.items():
    ms = sum(s.elapsed_time(e) for s, e, _ in evs)
    rows.append((ms / 16, len(evs) / 16, 1e3 * ms / len(evs), evs[0][2] / (1e6 * ms / len(evs)), k))
for ms, n, us, gbs, k in sorted(rows, key=lambda r: -r[0])[:40]:
    print(f'{ms:6.3f} ms/it {n:5.2f}/it {us:8.1f} us {gbs:8.1f} GB/s  {"tile   " if k[0] else "GENERIC"} x{k[1]} taps{k[2]} up{k[3]} down{k[4]} pad({k[5]},{k[6]}) -> {k[7]}x{k[8]} flip{k[9]}')
print('generic total %.3f ms/it' % sum(r[0] for r in rows if not r[4][0]))
