"""Per-launch timing of the hot kernels with HIP events on the launch stream.

bench.py installs a KernelTimer on the HIP backend; every conv / wgrad / upfirdn2d / bias_act
launch is then bracketed by two events recorded on the stream the kernel is launched on
(torch's current stream), and its ALGORITHMIC work is tallied next to it:

  conv / wgrad   flops = 2 * B * OC * IC * kh * kw * H_out * W_out   (transposed: ... * H_in * W_in)
  upfirdn2d      bytes = (numel_in + numel_out) * 4
  bias_act       bytes = (numel_in + numel_out) * 4 + C * 4 (+ B * inner * 4 with noise)
"""
import collections

import torch


_VARIANTS = {}
_MODES = {'f32': 0, 'bf16x3': 1, 'bf16': 2}


def conv_variant(geom, n_out, batch=1, k_in=64, mode='f32', in_hw=None):
    """Name of the kernel variant the library's dispatcher selects for this shape, asked of the dispatch code itself
    (gc_conv2d_variant_name: the launchers run in a no-launch probe mode), so the name cannot drift from the C++ side."""
    import ctypes
    from .. import _lib
    if in_hw is None:       # input extent from the output extent (any consistent value: the dispatch looks at the output side)
        in_hw = ((geom.out_h - 1) * geom.down + geom.kh - 2 * geom.pad_y, (geom.out_w - 1) * geom.down + geom.kw - 2 * geom.pad_x) if geom.up == 1 \
            else (-(-geom.out_h // geom.up), -(-geom.out_w // geom.up))
    key = (tuple(geom), n_out, batch, k_in, mode, tuple(in_hw))
    name = _VARIANTS.get(key)
    if name is None:
        desc = _lib.ConvDesc(batch, k_in, n_out, max(in_hw[0], 1), max(in_hw[1], 1), geom.out_h, geom.out_w, geom.kh, geom.kw, geom.up, geom.down, geom.pad_y, geom.pad_x)
        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.load().gc_conv2d_variant_name(desc, _MODES[mode], buf, 128), 'gc_conv2d_variant_name')
        name = _VARIANTS[key] = buf.value.decode()
    return name


def conv_flops(batch, k_in, n_out, in_h, in_w, geom):
    px = in_h * in_w if geom.up > 1 else geom.out_h * geom.out_w
    return 2.0 * batch * k_in * n_out * geom.kh * geom.kw * px


class KernelTimer:
    def __init__(self, only=None, names=None):
        self.records = []          # (name, start_event, end_event, work)
        self.enabled = True
        self.only = only           # None: every launch; else a tuple of kinds ('conv', 'wgrad', 'fir44', 'bias_act')
        self.names = names         # None: every kernel of those kinds; else the set of kernel names to bracket

    def start(self, kind=None, name=None):
        if not self.enabled or (self.only is not None and kind not in self.only):
            return None
        if self.names is not None and name is not None and name not in self.names:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def stop(self, name, start, work):
        if start is None:
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.records.append((name, start, e, work))

    def summary(self):
        """name -> dict(launches, total_ms, avg_us, work) ; call after torch.cuda.synchronize()."""
        acc = collections.OrderedDict()
        for name, s, e, work in self.records:
            d = acc.setdefault(name, {'launches': 0, 'total_ms': 0.0, 'work': 0.0})
            d['launches'] += 1
            d['total_ms'] += s.elapsed_time(e)
            d['work'] += work
        for d in acc.values():
            d['avg_us'] = 1e3 * d['total_ms'] / max(d['launches'], 1)
        return acc
