#!/usr/bin/env python3
"""Do an MFMA-bound kernel and an HBM-bound kernel overlap when they are launched on two streams?  (dev tool, GPU only)

In the backward pass the weight gradient of layer L is independent of the chain activation-backward -> input-gradient convolution ->
Blur adjoint that continues to layer L - 1.  This probe times pairs (A = weight gradient or convolution, B = streaming kernel) of the
FFHQ-1024 shapes: A alone, B alone, A then B on one stream, and A on a side stream while B runs on the main stream.
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch  # noqa: E402
from gan_control_amd.models.op import _backend  # noqa: E402
from gan_control_amd.models.op._backend import ConvGeom  # noqa: E402

be = _backend.get()
be.conv_mode = 'bf16x3'
dev = 'cuda'
side = torch.cuda.Stream()
REPS = 20


def wall(fn):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(REPS):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / REPS * 1e3


def pair(name, a, b):
    main = torch.cuda.current_stream()

    def both_serial():
        a(); b()

    def both_overlap():
        side.wait_stream(main)
        with torch.cuda.stream(side):
            a()
        b()
        main.wait_stream(side)

    ta, tb, ts, to = wall(a), wall(b), wall(both_serial), wall(both_overlap)
    print('%-64s A %7.1f us  B %7.1f us  serial %7.1f us  two streams %7.1f us  (saved %5.1f %% of serial; ideal max(A, B) = %7.1f)' %
          (name, ta, tb, ts, to, 100 * (ts - to) / ts, max(ta, tb)))


B = 4
k4 = torch.ones(4, 4, device=dev) / 16
for (ch, res) in [(512, 64), (256, 128), (128, 256), (64, 512), (32, 1024)]:
    g = ConvGeom(3, 3, 1, 1, 1, 1, res, res)
    x = torch.randn(B, ch, res, res, device=dev)
    dy = torch.randn(B, ch, res, res, device=dev)
    w = torch.randn(3, 3, ch, ch, device=dev)
    y = torch.randn(B, ch, res, res, device=dev)
    # the streaming work that follows in the main chain at the NEXT layer (same resolution in D's conv1 / G's second conv)
    pair(f'wgrad 3x3 {ch}->{ch} @{res}  ||  bias_act_bwd_reduce', lambda: be.conv2d_wgrad(x, dy, None, None, g), lambda: be.bias_act_bwd_reduce(dy, y, None, 0.2, 1.414))
    pair(f'wgrad 3x3 {ch}->{ch} @{res}  ||  input-gradient conv (MFMA || MFMA)', lambda: be.conv2d_wgrad(x, dy, None, None, g), lambda: be.conv2d(dy, w, None, None, g))
    xb = torch.randn(B, ch, res + 1, res + 1, device=dev)
    pair(f'conv 3x3 {ch}->{ch} @{res}  ||  fir44 blur {ch}x{res + 1}', lambda: be.conv2d(x, w, None, None, g), lambda: be.upfirdn2d(xb, k4, 1, 1, 1, 1, res, res, True))
    del x, dy, w, y, xb
