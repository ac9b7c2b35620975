import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd')); sys.path.insert(0, os.path.join(REPO, 'tests')); sys.path.insert(0, REPO)
from conftest import EmulatedBackend, rel_err
from gan_control_amd.models.op import _backend
from gan_control_amd.models.op._backend import ConvGeom
hip, emu = _backend.get(), EmulatedBackend()
hip.conv_mode = 'bf16x3'
for case in [(1, 64, 64, 33, 65, 3, 1, 1, 1), (1, 64, 64, 40, 64, 3, 1, 1, 1), (1, 64, 64, 40, 64, 1, 1, 1, 0), (2, 32, 32, 40, 70, 3, 1, 1, 1)]:
    b, K, N, h, w, k, up, down, pad = case
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(b, K, h, w, generator=gen); wt = torch.randn(k, k, K, N, generator=gen)
    oh, ow = h + 2 * pad - k + 1, w + 2 * pad - k + 1
    g = ConvGeom(k, k, up, down, pad, pad, oh, ow)
    ref = emu.conv2d(x.double(), wt.double(), None, None, g)
    out = hip.conv2d(x.cuda(), wt.cuda(), None, None, g).cpu()
    err = (out.double() - ref).abs()
    print(case, 'conv rel', rel_err(out, ref), 'interior', (err[:, :, 1:-1, 1:-1].max() / ref.abs().max()).item())
    dy = torch.randn(b, N, oh, ow, generator=gen)
    ref = emu.conv2d_wgrad(x.double(), dy.double(), None, None, g)
    out = hip.conv2d_wgrad(x.cuda(), dy.cuda(), None, None, g).cpu()
    print('      wgrad rel', rel_err(out, ref), 'per-tap', [(float((out[i, j].double() - ref[i, j]).abs().max() / ref.abs().max())) for i in range(k) for j in range(k)])
print('---- structure of the wgrad error, tap (0,0), case (1,64,64,40,64)')
b, K, N, h, w, k, up, down, pad = (1, 64, 64, 40, 64, 3, 1, 1, 1)
gen = torch.Generator().manual_seed(1)
x = torch.randn(b, K, h, w, generator=gen); dy = torch.randn(b, N, h, w, generator=gen)
g = ConvGeom(3, 3, 1, 1, 1, 1, h, w)
ref = emu.conv2d_wgrad(x.double(), dy.double(), None, None, g)
out = hip.conv2d_wgrad(x.cuda(), dy.cuda(), None, None, g).cpu().double()
e = (out - ref).abs()[0, 0]
print('bad k rows:', (e.max(1).values > 1e-3 * ref.abs().max()).nonzero().flatten().tolist())
print('bad n cols:', (e.max(0).values > 1e-3 * ref.abs().max()).nonzero().flatten().tolist()[:70])
# which image rows matter: zero x except one row, see which rows give wrong results
for row in [0, 1, 2, 3, 4, 5, 38, 39]:
    xz = torch.zeros_like(x); xz[:, :, row] = x[:, :, row]
    r2 = emu.conv2d_wgrad(xz.double(), dy.double(), None, None, g)
    o2 = hip.conv2d_wgrad(xz.cuda(), dy.cuda(), None, None, g).cpu().double()
    print('x row', row, 'err per ty', [float((o2[t] - r2[t]).abs().max() / ref.abs().max()) for t in range(3)])
for col in [0, 1, 7, 8, 30, 31, 32, 33, 62, 63]:
    xz = torch.zeros_like(x); xz[:, :, :, col] = x[:, :, :, col]
    r2 = emu.conv2d_wgrad(xz.double(), dy.double(), None, None, g)
    o2 = hip.conv2d_wgrad(xz.cuda(), dy.cuda(), None, None, g).cpu().double()
    print('x col', col, 'err per ty', [float((o2[t] - r2[t]).abs().max() / ref.abs().max()) for t in range(3)])
