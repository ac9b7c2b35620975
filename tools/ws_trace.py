#!/usr/bin/env python3
"""Dev tool (GPU): where does a tile's time go in conv_bf16x3_ws_kernel?  Needs a library built with -DGC_WS_TRACE=<workgroup + 1> (tools/build_alt.sh wstrace "-DGC_WS_TRACE=1"):
workgroup <n - 1> records s_memtime at the phase boundaries of its items for one multiplying wave (wave 0) and one staging wave (wave 8).
usage: GANCONTROL_HIP_LIB=.../libalt_wstrace.so python tools/ws_trace.py B K N res [k]"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from gan_control_amd import _lib
from gan_control_amd.models.op import _backend
from gan_control_amd.models.op._backend import ConvGeom
be = _backend.get(); be.conv_mode = 'bf16x3'
B, K, N, res = [int(v) for v in sys.argv[1:5]]
k = int(sys.argv[5]) if len(sys.argv) > 5 else 3
g = ConvGeom(k, k, 1, 1, k // 2, k // 2, res, res)
x = torch.randn(B, K, res, res, device='cuda'); w = torch.randn(k, k, K, N, device='cuda'); bias = torch.randn(N, device='cuda')
for _ in range(3):
    y = be.conv2d(x, w, None, None, g, epilogue=(bias, None, None, 0.2, 1.4, True))
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.library_path())
buf = (ctypes.c_ulonglong * 1024)()
assert lib.gc_debug_ws_trace(buf) == 0
MASK = (1 << 56) - 1
names = [{1: 'item top', 2: 'MFMAs done', 3: 'DMA + LDS waited', 4: 'barrier passed', 5: 'tile end: epilogue starts', 6: 'stores issued'},
         {1: 'interval top', 2: 'loads issued', 3: 'previous set landed', 4: 'converted + written', 5: 'barrier passed'}]
for role, label in ((0, 'multiplying wave 0'), (1, 'staging wave 8')):
    n = min(int(buf[role * 512 + 511]), 511)
    ev = [(int(buf[role * 512 + i]) >> 56, int(buf[role * 512 + i]) & MASK) for i in range(n)]
    print('== %s: %d events (s_memtime ticks, 100 MHz constant clock -> x10 ns)' % (label, n))
    t0 = ev[0][1]
    # per-phase totals over the recorded window
    tot = {}
    for (ta, a), (tb, b_) in zip(ev, ev[1:]):
        key = '%s -> %s' % (names[role].get(ta, ta), names[role].get(tb, tb))
        d = tot.setdefault(key, [0, 0]); d[0] += b_ - a; d[1] += 1
    span = ev[-1][1] - t0
    for key, (d, c) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
        print('  %-58s %8d ticks  %5.1f %%  (%d x, avg %.1f)' % (key, d, 100.0 * d / span, c, d / c))
    print('  first 40 events:', ' '.join('%d@%d' % (t, tm - t0) for t, tm in ev[:40]))
