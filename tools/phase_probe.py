import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from gan_control_amd.models.op import _backend
from gan_control_amd.models.op._backend import ConvGeom
be = _backend.get(); be.conv_mode = 'bf16x3'
for B, K, N, res in [(8, 32, 32, 1024), (8, 64, 64, 512), (4, 512, 512, 64)]:
    g = ConvGeom(3, 3, 1, 1, 1, 1, res, res)
    x = torch.randn(B, K, res, res, device='cuda'); w = torch.randn(3, 3, K, N, device='cuda')
    for _ in range(3):
        y = be.conv2d(x, w, None, None, g)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); y = be.conv2d(x, w, None, None, g); e.record(); torch.cuda.synchronize()
    d = y[0, 0, 0, :9].cpu().tolist()
    print(f'{K}->{N}@{res} B{B}: kernel {s.elapsed_time(e)*1e3:.0f} us; ticks: issue {d[1]:.0f} mfma {d[2]-d[1]:.0f} bar1 {d[3]-d[2]:.0f} commit {d[4]-d[3]:.0f} store {d[5]-d[4]:.0f} bar2 {d[6]-d[5]:.0f}  | item total {d[6]:.0f}; block total {d[7]:.0f} for {d[8]:.0f} items')
