#!/usr/bin/env python3
"""Per-shape times and achieved GB/s of every HBM-bound launch of the step -- FIR variants, activation passes, reductions -- over 16 training
iterations (one full lazy-regulariser period): the twin of tools/shape_profile.py for the memory-bound families (dev tool, GPU only).
Bytes are the algorithmic ones: every input tensor read once, every output written once."""
import collections, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from gan_control_amd.models.op import _backend
from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
be = _backend.get()
be.conv_mode = 'bf16x3'
tr = GeneratorTrainer(default_config(size, batch), device='cuda', seed=0)
real = tr.synthetic_batch()
for it in range(4):
    tr.train_iteration(it, real)
rec = collections.defaultdict(list)
NAMES = ['upfirdn2d', 'upfirdn2d_act', 'upfirdn2d_mask', 'upfirdn2d_actbwd', 'bias_act', 'bias_act_bwd', 'bias_act_bwd_reduce', 'bias_act_bwd_reduce_adjoint',
         'plane_dot', 'channel_sum', 'rows_sum_div', 'pw_act_wgrad', 'pw_act_dgrad']


def tensors(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, (tuple, list)):
        for o in obj:
            yield from tensors(o)


def wrap(name, fn):
    def inner(*a, **kw):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn(*a, **kw)
        e.record()
        ins = [t for t in tensors(list(a) + list(kw.values())) if t.is_cuda]
        outs = [t for t in tensors(out) if t.is_cuda]
        by = 4.0 * (sum(t.numel() for t in ins) + sum(t.numel() for t in outs))
        big = max(ins + outs, key=lambda t: t.numel())
        extra = ''
        if name.startswith('upfirdn2d'):
            o = outs[0]
            extra = ' -> %dx%d' % (o.shape[2], o.shape[3])
        rec[(name, tuple(big.shape), extra, len(ins))].append((s, e, by))
        return out
    return inner


for n in NAMES:
    if hasattr(be, n):
        setattr(be, n, wrap(n, getattr(be, n)))
iters = 16
for it in range(16, 16 + iters):
    tr.train_iteration(it, real)
torch.cuda.synchronize()
rows = []
for key, evs in rec.items():
    ms = sum(s.elapsed_time(e) for s, e, _ in evs)
    rows.append((ms / iters, len(evs) / iters, 1e3 * ms / len(evs), evs[0][2] / (1e6 * ms / len(evs)), key))
tot = sum(r[0] for r in rows)
print(f'total {tot:.2f} ms/iter in {sum(r[1] for r in rows):.0f} launches/iter of the memory-bound families')
fam = collections.defaultdict(lambda: [0.0, 0.0])
for ms, n, us, gbs, key in rows:
    fam[key[0]][0] += ms; fam[key[0]][1] += n
for k, (ms, n) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print(f'  {k:30s} {ms:6.2f} ms/it {n:6.1f} launches/it')
for ms, n, us, gbs, key in sorted(rows, key=lambda r: -r[0]):
    name, shape, extra, nin = key
    print(f'{ms:6.3f} ms/it {n:5.2f}/it {us:8.1f} us {gbs:7.0f} GB/s  {name:28s} {str(list(shape)):24s}{extra} ({nin} inputs)')
