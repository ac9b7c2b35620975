#!/usr/bin/env python3
"""Dev tool (GPU): race screen of the wave-specialised kernels.  LDS-DMA data is ordered for its readers only by the issuing wave's
vmcnt wait + a barrier; a misplaced read passes single checks whenever the DMA happens to land first.  The same launch is repeated many
times, on a busy device, and every output must be bit-identical to the first one -- and to the one-role kernel's output when
GANCONTROL_HIP_LIB_REF points at a build with -DGC_NO_WS (same summation order)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from gan_control_amd.models.op import _backend
from gan_control_amd.models.op._backend import ConvGeom
be = _backend.get(); be.conv_mode = 'bf16x3'
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
gen = torch.Generator().manual_seed(1)
bad = 0
for (B, K, N, res, k) in [(4, 512, 512, 64, 3), (8, 256, 256, 128, 3), (4, 128, 128, 256, 3), (4, 64, 64, 512, 3), (4, 32, 32, 1024, 3), (4, 128, 256, 128, 1), (2, 80, 64, 257, 3)]:
    x = torch.randn(B, K, res, res, generator=gen).cuda(); w = torch.randn(k, k, K, N, generator=gen).cuda()
    si = torch.randn(B, K, generator=gen).cuda(); so = (torch.rand(B, N, generator=gen) + 0.5).cuda()
    bias = torch.randn(N, generator=gen).cuda(); nz = torch.randn(B, 1, res, res, generator=gen).cuda(); nw = torch.randn(1, generator=gen).cuda()
    g = ConvGeom(k, k, 1, 1, k // 2, k // 2, res, res)
    ep = (bias, nz, nw, 0.2, 1.4, True)
    first = be.conv2d(x, w, si, so, g, epilogue=ep).clone()
    junk = torch.randn(64 << 20, device='cuda')
    n_bad = 0
    for i in range(reps):
        if i % 3 == 0:
            junk.mul_(1.0001)                   # other traffic in flight next to the launch
        y = be.conv2d(x, w, si, so, g, epilogue=ep)
        if not torch.equal(y, first):
            n_bad += 1
    torch.cuda.synchronize()
    print(f'{B}x{K}->{N} @{res} k{k}: {reps} launches, {n_bad} differ from the first; checksum {float(first.double().sum()):.6f} finite={bool(torch.isfinite(first).all())}')
    bad += n_bad
# The weight-gradient kernels (round 4): deterministic two-stage reductions over pixel splits, no atomics -- every launch must reproduce the
# first bit for bit, stride 1 and stride 2, with and without per-sample scales, and the per-sample form (dw + its B shares).
for (B, K, N, res, k, down) in [(4, 512, 512, 64, 3, 1), (8, 256, 256, 128, 3, 1), (4, 32, 32, 1024, 3, 1), (8, 128, 256, 257, 3, 2), (8, 32, 64, 1025, 3, 2), (4, 128, 256, 128, 1, 1)]:
    pad = k // 2 if down == 1 else 0
    oh = (res + 2 * pad - k) // down + 1
    g = ConvGeom(k, k, 1, down, pad, pad, oh, oh)
    x = torch.randn(B, K, res, res, generator=gen).cuda(); dy = torch.randn(B, N, oh, oh, generator=gen).cuda()
    si = torch.randn(B, K, generator=gen).cuda(); so = (torch.rand(B, N, generator=gen) + 0.5).cuda()
    for scales in ((None, None), (si, so)):
        first = be.conv2d_wgrad(x, dy, scales[0], scales[1], g).clone()
        n_bad = 0
        wreps = max(reps // 5, 20)
        for i in range(wreps):
            if i % 3 == 0:
                junk.mul_(1.0001)
            if not torch.equal(be.conv2d_wgrad(x, dy, scales[0], scales[1], g), first):
                n_bad += 1
        torch.cuda.synchronize()
        print(f'wgrad {B}x{K}->{N} @{res} k{k} down{down} scales={scales[0] is not None}: {wreps} launches, {n_bad} differ from the first; finite={bool(torch.isfinite(first).all())}')
        bad += n_bad
    del x, dy
# Round 5: the transposed layers -- split over K (partial sums + finish pass), H x W main region + edge kernel (two launches writing disjoint parts of one
# tensor; the edge kernel adds its four channel quarters through LDS in a fixed order), the zero-stuffed small-plane form -- and the small-plane weight gradient.
for (B, K, N, res) in [(4, 512, 512, 8), (4, 512, 512, 16), (8, 512, 512, 32), (4, 512, 256, 64), (4, 256, 128, 128), (8, 512, 512, 4)]:
    oh = 2 * res + 1
    g = ConvGeom(3, 3, 2, 1, 2, 2, oh, oh)
    x = torch.randn(B, K, res, res, generator=gen).cuda(); w = torch.randn(3, 3, K, N, generator=gen).cuda()
    si = torch.randn(B, K, generator=gen).cuda(); so = (torch.rand(B, N, generator=gen) + 0.5).cuda()
    first = be.conv2d(x, w, si, so, g).clone()
    n_bad = 0
    treps = max(reps // 2, 20)
    for i in range(treps):
        if i % 3 == 0:
            junk.mul_(1.0001)
        if not torch.equal(be.conv2d(x, w, si, so, g), first):
            n_bad += 1
    torch.cuda.synchronize()
    print(f'convT {B}x{K}->{N} @{res}: {treps} launches, {n_bad} differ from the first; finite={bool(torch.isfinite(first).all())}')
    bad += n_bad
for (B, K, N, res, down) in [(8, 512, 512, 8, 1), (4, 512, 512, 17, 2), (8, 513, 512, 4, 1)]:
    pad = 1 if down == 1 else 0
    oh = (res + 2 * pad - 3) // down + 1
    g = ConvGeom(3, 3, 1, down, pad, pad, oh, oh)
    x = torch.randn(B, K, res, res, generator=gen).cuda(); dy = torch.randn(B, N, oh, oh, generator=gen).cuda()
    first = be.conv2d_wgrad(x, dy, None, None, g).clone()
    n_bad = sum(0 if torch.equal(be.conv2d_wgrad(x, dy, None, None, g), first) else 1 for _ in range(max(reps // 2, 20)))
    torch.cuda.synchronize()
    print(f'small wgrad {B}x{K}->{N} @{res} down{down}: {max(reps // 2, 20)} launches, {n_bad} differ from the first; finite={bool(torch.isfinite(first).all())}')
    bad += n_bad
# Round 6: the wave-specialised stride-2 kernel (conv_s2ws.hip): LDS-DMA weight slabs behind counted vmcnt waits (a finished tile's 64 stores per lane may stay in
# flight across the barrier), E / O half stages filled by the staging waves one sub-item ahead -- both output-block widths, full epilogue with a residual,
# a pitched-free dense input and a ragged plane; additionally the result must equal the one-role kernel's to rounding (different summation order: not bit for bit).
from gan_control_amd.utils.profiling import conv_variant
for (B, K, N, res) in [(8, 128, 256, 257), (8, 32, 64, 1025), (4, 256, 512, 129), (4, 64, 128, 513), (3, 48, 192, 259)]:
    oh = (res - 3) // 2 + 1
    g = ConvGeom(3, 3, 1, 2, 0, 0, oh, oh)
    assert conv_variant(g, N, B, K, 'bf16x3', (res, res)).startswith('conv_s2ws_bf16x3_kernel'), 'shape does not reach the stride-2 ws kernel'
    x = torch.randn(B, K, res, res, generator=gen).cuda(); w = torch.randn(3, 3, K, N, generator=gen).cuda()
    si = torch.randn(B, K, generator=gen).cuda(); so = (torch.rand(B, N, generator=gen) + 0.5).cuda()
    bias = torch.randn(N, generator=gen).cuda(); resid = torch.randn(B, N, oh, oh, generator=gen).cuda()
    ep = (bias, None, None, 0.2, 1.4, True, resid)
    first = be.conv2d(x, w, si, so, g, epilogue=ep).clone()
    ref = torch.nn.functional.conv2d((x * si[:, :, None, None]).double(), w.permute(3, 2, 0, 1).double(), stride=2) * so[:, :, None, None].double() + bias.double()[None, :, None, None]
    ref = torch.where(ref > 0, ref, 0.2 * ref) * 1.4 + resid.double()
    err = float((first.double() - ref).norm() / ref.norm())
    n_bad = 0
    sreps = max(reps // 2, 20)
    for i in range(sreps):
        if i % 3 == 0:
            junk.mul_(1.0001)
        if not torch.equal(be.conv2d(x, w, si, so, g, epilogue=ep), first):
            n_bad += 1
    torch.cuda.synchronize()
    print(f'conv s2ws {B}x{K}->{N} @{res}: {sreps} launches, {n_bad} differ from the first; relative error vs float64 {err:.2e}; finite={bool(torch.isfinite(first).all())}')
    bad += n_bad + (err > 5e-5)
    del x, w, resid, ref
print('RACE SCREEN', 'FAILED' if bad else 'clean')
