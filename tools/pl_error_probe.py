import sys, os, torch, math
sys.path[:0]=['/root/repo','/root/repo/gan-control_amd','/root/repo/tests']
import conftest, step_checks, op_checks as oc
from conftest import load_golden, EmulatedBackend
from gan_control_amd.models.op import _backend, modulated_conv as mc
from gan_control_amd.models.gan_model import Generator
emu=EmulatedBackend()
_backend._install_for_tests(emu)
s=load_golden('step'); t=lambda k: torch.from_numpy(s[k])
seeds=[int(v) for v in s['noise_seeds']]
nz=oc.seeded_noise(32,2,seeds[2])
rec={}
def wrap(name, fn):
    def f(*a, **k):
        out=fn(*a, **k)
        rec.setdefault(cur[0],[]).append((name, out if not isinstance(out, tuple) else out[0]))
        return out
    return f
for nm in ('plane_dot','conv2d','bias_act_bwd_reduce','conv2d_wgrad'):
    setattr(emu, nm, wrap(nm, getattr(emu, nm)))
cur=['x']
res={}
for dt in (torch.float64, torch.float32):
    cur[0]=str(dt)
    tr=step_checks.make_trainer('cpu'); tr.generator.to(dt)
    for p in tr.generator.parameters(): p.requires_grad_(True)
    img, lat = tr.generator([t('z_pl').to(dt)], noise=[m.to(dt) for m in nz], return_latents=True)
    rec.setdefault(cur[0],[]).append(('MARK',img))
    gr = Generator.g_path_regularize_grad(img, lat, pl_noise=t('pl_noise').to(dt))
a=rec[str(torch.float64)]; b=rec[str(torch.float32)]
assert len(a)==len(b)
for (n1,x),(n2,y) in zip(a,b):
    print(n1, tuple(x.shape), '%.2e'%float((y.double()-x).norm()/x.norm().clamp_min(1e-300)))
