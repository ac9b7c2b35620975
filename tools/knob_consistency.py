import os, sys, subprocess, json
REPO = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
code = r'''
import os, sys, torch
sys.path.insert(0, os.path.join(%r, 'gan-control_amd'))
from gan_control_amd.models.op import _backend
from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config
_backend.get().conv_mode = os.environ.get('KNOB_MODE', 'f32')
tr = GeneratorTrainer(default_config(256, 4), device='cuda', seed=0, fused_adam=False)
real = tr.synthetic_batch()
import random; random.seed(0)
for i in range(9):
    tr.train_iteration(i, real)
torch.cuda.synchronize()
sd = {k: v.detach().double().cpu() for k, v in list(tr.generator.state_dict().items()) + [('d.' + k, v) for k, v in tr.discriminator.state_dict().items()] if v.dtype.is_floating_point}
torch.save({'sd': sd, 'stats': {k: float(v) for k, v in tr.stats.items() if torch.is_tensor(v) and v.numel() == 1}}, sys.argv[1])
''' % REPO
outs = {}
OFF = {'GANCONTROL_FUSE_BLUR_ADJOINT': '0', 'GANCONTROL_FUSE_ACT_BWD_BLUR': '0', 'GANCONTROL_FUSED_STYLE': '0', 'GANCONTROL_WEIGHT_BATCH': '0',
       'GANCONTROL_FUSE_PW_ACT': '0', 'GANCONTROL_WGRAD_SAMPLES': '0', 'GANCONTROL_FORK_TORGB': '0'}
for tag, env in (('on', {}), ('off', OFF), ('on_bf16x3', {'KNOB_MODE': 'bf16x3'}), ('off_bf16x3', dict(OFF, KNOB_MODE='bf16x3'))):
    path = '/tmp/knob_%s.pt' % tag
    subprocess.run([sys.executable, '-c', code, path], check=True, env=dict(os.environ, **env))
    outs[tag] = path
import torch
runs = {k: torch.load(v) for k, v in outs.items()}


def compare(x, y):
    a, b = runs[x], runs[y]
    worst = max((v - b['sd'][k]).abs().max().item() for k, v in a['sd'].items())
    big = sum(int(((v - b['sd'][k]).abs() > 1e-3).sum()) for k, v in a['sd'].items())
    n = sum(v.numel() for v in a['sd'].values())
    print('%-12s vs %-12s: max |param difference| %.3e, %.2f %% of %d elements differ by more than 1e-3; g_adv_loss %.4f vs %.4f' %
          (x, y, worst, 100.0 * big / n, n, a['stats']['g_adv_loss'], b['stats']['g_adv_loss']))


print('9 iterations at 256 x 256, batch 4, same seeds (first Adam steps are sign-like: any rounding difference near a zero gradient flips a step of 2 lr)')
compare('on', 'off')                 # the round-3 fusions against the round-2 paths, exact fp32 convolutions
compare('on', 'on_bf16x3')           # the natural scale of that divergence: the same code in the two convolution arithmetics (~1e-5 apart per layer)
compare('off', 'off_bf16x3')
compare('on_bf16x3', 'off_bf16x3')
