#!/usr/bin/env python3
"""Where is the GPU idle inside a training iteration?  (dev tool, GPU only)

Profiles a few iterations with torch.profiler, merges the device-side kernel intervals and reports busy / idle time, the
distribution of the idle gaps and, for the largest gaps, the kernels on either side.
usage: gap_profile.py [--size 1024] [--batch 4] [--iters 4] [--first 17]"""
import argparse, os, sys, json, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from torch.profiler import profile, ProfilerActivity
from gan_control_amd.models.op import _backend
from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config

ap = argparse.ArgumentParser()
ap.add_argument('--size', type=int, default=1024); ap.add_argument('--batch', type=int, default=4)
ap.add_argument('--iters', type=int, default=4); ap.add_argument('--first', type=int, default=17)
ap.add_argument('--precision', default='bf16x3')
a = ap.parse_args()
_backend.get().conv_mode = a.precision
tr = GeneratorTrainer(default_config(a.size, a.batch), device='cuda', seed=0)
real = tr.synthetic_batch()
for i in range(17):
    tr.train_iteration(i, real)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for i in range(a.first, a.first + a.iters):
        tr.train_iteration(i, real)
    torch.cuda.synchronize()
path = os.path.join(tempfile.gettempdir(), 'gap_trace.json')
prof.export_chrome_trace(path)
ev = json.load(open(path))['traceEvents']
kern = sorted((e['ts'], e['ts'] + e['dur'], e['name']) for e in ev if e.get('cat') in ('kernel', 'gpu_memcpy', 'gpu_memset') and 'dur' in e)
t0, t1 = kern[0][0], max(k[1] for k in kern)
busy, gaps, end, last = 0.0, [], kern[0][0], None
for s, e, n in kern:
    if s > end:
        gaps.append((s - end, last, n))
        busy += e - s
        end = e
    else:
        busy += max(0.0, e - max(s, end))
        end = max(end, e)
    last = n if e >= end else last
wall = t1 - t0
print(f'{len(kern)} device activities over {wall / 1e3:.1f} ms ({wall / 1e3 / a.iters:.1f} ms / iteration): busy {busy / 1e3:.1f} ms ({100 * busy / wall:.1f} %), idle {(wall - busy) / 1e3:.1f} ms')
for lo, hi in ((0, 5), (5, 10), (10, 20), (20, 50), (50, 200), (200, 1000), (1000, 1e9)):
    g = [x[0] for x in gaps if lo <= x[0] < hi]
    print(f'  gaps {lo:5.0f}..{hi:<6.0f} us: {len(g):6d}  total {sum(g) / 1e3:8.2f} ms')
print('largest gaps (us, kernel before -> kernel after):')
for g, b, n in sorted(gaps, reverse=True)[:25]:
    print(f'  {g:8.1f}  {str(b)[:70]} -> {n[:70]}')
