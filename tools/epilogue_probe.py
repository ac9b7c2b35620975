#!/usr/bin/env python3
"""Dev tool (GPU): the wave-specialised stride-1 kernel with each of its epilogue forms at the FFHQ-1024 layer shapes (B = 4): bare stores,
out_scale only (the input-gradient launches), the full fused noise + bias + leaky-ReLU epilogue with per-sample scales (the forward launches
of G) and bias + activation without scales (D), microseconds and TFLOP/s.  Run it once per library (GANCONTROL_HIP_LIB) for a same-box A/B;
prints a checksum per case so that two builds can also be compared for equal results."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch  # noqa: E402
from gan_control_amd.models.op import _backend  # noqa: E402
from gan_control_amd.models.op._backend import ConvGeom  # noqa: E402


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


def main():
    be = _backend.get()
    be.conv_mode = 'bf16x3'
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    torch.manual_seed(0)
    for ch, res in [(512, 64), (256, 128), (128, 256), (64, 512), (32, 1024)]:
        g = ConvGeom(3, 3, 1, 1, 1, 1, res, res)
        x = torch.randn(B, ch, res, res, device='cuda'); w = torch.randn(3, 3, ch, ch, device='cuda') / (3 * ch ** 0.5)
        si, so = torch.rand(B, ch, device='cuda') + 0.5, torch.rand(B, ch, device='cuda') + 0.5
        bias, nz, nw = torch.randn(ch, device='cuda'), torch.randn(B, 1, res, res, device='cuda'), torch.randn(1, device='cuda')
        resid = torch.randn(B, ch, res, res, device='cuda')
        fl = 2.0 * B * ch * ch * 9 * res * res
        cases = [('bare', None, None, None),
                 ('out_scale', si, so, None),
                 ('scale+residual', None, so, (None, None, None, 1.0, 1.0, 0, resid)),
                 ('G fwd: scales+noise+bias+act', si, so, (bias, nz, nw, 0.2, 2 ** 0.5, 1)),
                 ('D fwd: bias+act', None, None, (bias, None, None, 0.2, 2 ** 0.5, 1)),
                 ('D fwd: bias+act+residual', None, None, (bias, None, None, 0.2, 2 ** 0.5, 1, resid))]
        for name, a, b, ep in cases:
            y = be.conv2d(x, w, a, b, g, epilogue=ep)
            t = timeit(lambda: be.conv2d(x, w, a, b, g, epilogue=ep))
            print('%4d -> %4d @%4d  %-30s %8.1f us %7.1f TF/s   sum %.6e' % (ch, ch, res, name, t * 1e6, fl / t / 1e12, float(y.double().sum())))
        del x, w, nz, resid


if __name__ == '__main__':
    main()
