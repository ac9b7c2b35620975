#!/usr/bin/env python3
"""Where does the HOST time of a phase go?  (dev tool, GPU only)  cProfile over a few repetitions of one phase of the iteration
(d_step | r1 | g_step | pl), without device synchronisation inside the profiled region: top functions by own time."""
import cProfile
import os
import pstats
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch  # noqa: E402
from gan_control_amd.models.op import _backend  # noqa: E402
from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config  # noqa: E402
from gan_control_amd.trainers.utils import requires_grad, make_mini_batch_from_noise  # noqa: E402

phase = sys.argv[1] if len(sys.argv) > 1 else 'pl'
if len(sys.argv) > 2 and sys.argv[2] == 'inline':       # run backward on the calling thread so that cProfile sees the backward functions too
    torch.autograd.set_multithreading_enabled(False)
_backend.get().conv_mode = 'bf16x3'
size, batch = 1024, 4
tr = GeneratorTrainer(default_config(size, batch), device='cuda', seed=0)
real = tr.synthetic_batch()
for i in range(2):
    tr.train_iteration(i * 16, real)


_it = [1]


def run():
    if phase == 'iter':          # a plain iteration (no lazy regulariser): i = 1, 2, 3, 5, ...
        while _it[0] % 4 == 0:
            _it[0] += 1
        tr.train_iteration(_it[0], real)
        _it[0] += 1
        return
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
prof = cProfile.Profile()
n = 5
for _ in range(n):
    torch.cuda.synchronize()
    prof.enable()
    run()
    prof.disable()
torch.cuda.synchronize()
st = pstats.Stats(prof)
st.sort_stats('tottime')
print('phase', phase, ': host time per repetition %.2f ms' % (st.total_tt / n * 1e3))
st.print_stats(int(os.environ.get('TOP', '28')))
