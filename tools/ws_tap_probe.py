#!/usr/bin/env python3
"""Dev tool (GPU): what do four staging waves sustain?  The wave-specialised forward kernel with ONE tap per staged 16-channel chunk (its 1 x 1
instantiation) next to the 3 x 3 form, at the FFHQ-1024 channel counts.  A 1 x 1 item stages ~1 000 patch units for 12 MFMAs per multiplying wave,
so its time per item (1.7 us measured) is the staging pipeline's own: the price list for any kernel that feeds fewer taps per staged unit than a
stride-1 3 x 3 layer does (a phase-plane stride-2 kernel: 2.25 taps per chunk on average -- DESIGN.md section 6, round 4)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from gan_control_amd.models.op import _backend
from gan_control_amd.models.op._backend import ConvGeom
from gan_control_amd.utils.profiling import conv_variant
be = _backend.get(); be.conv_mode = 'bf16x3'
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3
for B in (4, 8):
    for c, res in [(512, 64), (256, 128), (128, 256), (64, 512)]:
        for k in (1, 3):
            g = ConvGeom(k, k, 1, 1, k // 2, k // 2, res, res)
            x = torch.randn(B, c, res, res, device='cuda'); w = torch.randn(k, k, c, c, device='cuda')
            t = timeit(lambda: be.conv2d(x, w, None, None, g))
            fl = 2.0 * B * c * c * k * k * res * res
            print(B, c, res, 'k%d' % k, conv_variant(g, c, B, c, 'bf16x3', (res, res)), '%.1f us %.1f TF/s' % (t * 1e6, fl / t / 1e12))
