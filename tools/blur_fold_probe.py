#!/usr/bin/env python3
"""What would folding the Blur into the neighbouring convolution cost?  (dev tool, GPU only)

G: transposed 3x3 (stride 2) followed by the 4x4 Blur == a transposed 6x6 convolution == FOUR 3x3 stride-1 convolutions of the input, one per
output phase (gan_model.py:295-307).  D: Blur followed by a 3x3 stride-2 convolution == a 6x6 stride-2 convolution == four times the multiply-
adds of the 3x3 one (:857-872).  A folded kernel saves the write + read of the (2H + 1)^2 / (H + 1)^2 intermediate and pays 4x the matrix work.
This probe times, per layer of the FFHQ-1024 networks, today's two launches against a LOWER BOUND of the folded form built from the fastest
existing kernels: four launches of the 3x3 convolution that one phase amounts to (no interleaved store, no epilogue, input re-read from cache).
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
dev, B, REPS = 'cuda', 4, 10
k4 = (torch.tensor([1., 3., 3., 1.])[:, None] * torch.tensor([1., 3., 3., 1.])[None, :] / 64 * 4).to(dev)


def wall(fn):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(REPS):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / REPS * 1e3


print('G: transposed 3x3 + Blur(+noise + bias + leaky-ReLU)   vs   four 3x3 stride-1 phase convolutions (lower bound of a folded kernel)')
for ic, oc, res in [(512, 256, 64), (256, 128, 128), (128, 64, 256), (64, 32, 512)]:
    x = torch.randn(B, ic, res, res, device=dev)
    w = torch.randn(3, 3, ic, oc, device=dev)
    si, so = torch.rand(B, ic, device=dev) + 0.5, torch.rand(B, oc, device=dev) + 0.5
    gt = ConvGeom(3, 3, 2, 1, 2, 2, 2 * res + 1, 2 * res + 1)
    g1 = ConvGeom(3, 3, 1, 1, 1, 1, res, res)
    bias, nz, nw = torch.randn(oc, device=dev), torch.randn(B, 1, 2 * res, 2 * res, device=dev), torch.randn(1, device=dev)
    mid = be.conv2d(x, w, si, so, gt)
    t_ct = wall(lambda: be.conv2d(x, w, si, so, gt))
    t_fir = wall(lambda: be.upfirdn2d_act(mid, k4, 1, 1, 2 * res, 2 * res, True, bias, nz, nw, 0.2, 1.414))
    t_ph = wall(lambda: be.conv2d(x, w, si, so, g1))
    print('  %4d -> %4d @%4d^2: today %6.1f + %6.1f = %6.1f us   folded >= 4 x %6.1f = %6.1f us   (%+5.1f %%)' %
          (ic, oc, res, t_ct, t_fir, t_ct + t_fir, t_ph, 4 * t_ph, 100 * (4 * t_ph / (t_ct + t_fir) - 1)))
    del x, w, mid, nz

print('D: Blur + 3x3 stride-2   vs   four 3x3 stride-2 convolutions (the multiply-adds of the 6x6 stride-2 form)')
for ic, oc, res in [(32, 64, 1024), (64, 128, 512), (128, 256, 256), (256, 512, 128)]:
    x = torch.randn(2 * B, ic, res, res, device=dev)            # the discriminator step sees fake and real together
    w = torch.randn(3, 3, ic, oc, device=dev)
    g2 = ConvGeom(3, 3, 1, 2, 0, 0, res // 2, res // 2)
    xb = be.upfirdn2d(x, k4, 1, 1, 2, 2, res + 1, res + 1, True)
    t_fir = wall(lambda: be.upfirdn2d(x, k4, 1, 1, 2, 2, res + 1, res + 1, True))
    t_s2 = wall(lambda: be.conv2d(xb, w, None, None, g2))
    print('  %4d -> %4d @%4d^2: today %6.1f + %6.1f = %6.1f us   folded >= 4 x %6.1f = %6.1f us   (%+5.1f %%)' %
          (ic, oc, res, t_fir, t_s2, t_fir + t_s2, t_s2, 4 * t_s2, 100 * (4 * t_s2 / (t_fir + t_s2) - 1)))
    del x, w, xb
