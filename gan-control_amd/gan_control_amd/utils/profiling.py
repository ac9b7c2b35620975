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


def conv_variant(geom, n_out, batch=1, k_in=64, mode='f32'):
    """Name of the conv_mfma_kernel tile configuration gc_conv2d_f32 selects (csrc/conv.hip, dispatch_conv).

    Template arguments: <WG_OC, WG_PX, KSPLIT, WOC, WPX, TPW>; the (up, down, taps) triple follows.  In bf16x3 mode
    the same shapes run on conv_bf16x3_kernel (planes wider than 8 px and >= 16 input channels).
    """
    qw, qh = -(-geom.out_w // geom.up), -(-geom.out_h // geom.up)
    geo = '|up%d,down%d,k%d' % (geom.up, geom.down, geom.kh)
    if geom.kh == 1 and geom.up == 1 and geom.down == 1 and geom.pad_y == 0 and geom.pad_x == 0 and (k_in <= 4 or n_out <= 4) \
            and geom.out_h * geom.out_w * batch >= 1 << 18:
        return ('pw_narrow_kernel' if n_out <= 4 else 'pw_widen_kernel') + geo      # csrc/pointwise.hip
    if mode in ('bf16x3', 'bf16') and 16 <= k_in <= 1024 and qw > 8:
        if geom.up == 2 and geom.kh == 3 and geom.pad_y == 2 and geom.pad_x == 2:
            # convt_fused_bf16x3_kernel<WG_OC, WG_PX, WPX, TPW, EPI> (dispatch_t): the four output phases in one workgroup
            tq = -(-geom.out_w // 2)
            tpw = 16 if -(-tq // 16) * 16 < -(-tq // 32) * 32 else 32
            return ('convt_fused_bf16x3_kernel<1,4,2,%d>' % tpw if n_out <= 32 else 'convt_fused_bf16x3_kernel<2,2,2,%d>' % tpw) + geo
        # conv_bf16x3_kernel<WG_OC, WG_PX, WOC, WPX> (csrc/conv_bf16x3.hip, dispatch)
        if geom.down == 2:
            return ('conv_bf16x3_kernel<1,4,1,1>' if n_out <= 32 else 'conv_bf16x3_kernel<1,4,2,1>') + geo
        if n_out <= 32:
            return 'conv_bf16x3_kernel<1,4,1,2>' + geo
        big = -(-qw // 32) * -(-qh // 8) * geom.up * geom.up * batch * -(-n_out // 64)
        return ('conv_bf16x3_kernel<1,4,2,1>' if big < 512 else 'conv_bf16x3_kernel<1,4,2,2>') + geo
    if qw <= 16:
        tpw = 4 if qw <= 4 else (8 if qw <= 8 else 16)
        return 'conv_mfma_kernel<1,1,4,32,1,1,%d>' % tpw + geo
    if geom.down == 2:
        if n_out <= 64:
            return 'conv_mfma_kernel<2,2,1,1,1,32>' + geo
    else:
        if n_out <= 32:
            return 'conv_mfma_kernel<1,4,1,1,4,32>' + geo
        if n_out <= 64:
            return 'conv_mfma_kernel<1,4,1,2,2,32>' + geo
    big = -(-qw // 32) * -(-qh // 4) * geom.up * geom.up * batch * -(-n_out // 128)
    if big < 512:
        return 'conv_mfma_kernel<2,2,1,1,1,32>' + geo
    return 'conv_mfma_kernel<2,2,1,2,2,32>' + geo


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
