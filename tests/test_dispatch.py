"""Which kernel takes which layer of the FFHQ-1024 step (no GPU needed: gc_conv2d_variant_name runs the dispatch code in a no-launch probe mode).

The launchers decide by shape -- tile counts, LDS fit, workgroups per CU -- and several of those rules were MEASURED (DESIGN.md section 8, items 6 and 7:
a transposed layer on the H x W + edge form below 512 main-region workgroups is slower, more K slices than ~512 workgroups need are slower, ...).  These
tests pin the decisions for the shapes the training step launches, so that a change of one eligibility rule shows up here and not only as a slower bench line.
"""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))

from gan_control_amd import _lib                                        # noqa: E402
from gan_control_amd.models.op._backend import ConvGeom                 # noqa: E402
from gan_control_amd.utils.profiling import conv_variant                # noqa: E402


def _out(n, k, up, down, pad):
    return (n - 1) * up + k - 2 * (k - 1 - pad) if up > 1 else (n + 2 * pad - k) // down + 1


def _probe(b, K, N, h, k, up, down, pad, mode='bf16x3'):
    o = _out(h, k, up, down, pad)
    geom = ConvGeom(k, k, up, down, pad, pad, o, o)
    desc = _lib.ConvDesc(b, K, N, h, h, o, o, k, k, up, down, pad, pad)
    return conv_variant(geom, N, b, K, mode, (h, h)), int(_lib.load().gc_conv2d_bf16x3_splitk_bytes(desc))


# (batch, K, N, plane, taps, up, down, pad) -> (kernel name prefix, split over K?)
STEP_SHAPES = [
    # stride 1: wave-specialised from 64^2 up, one-role kernel with K slices below, the exact-fp32 small-plane kernel at <= 8 x 8
    ((4, 512, 512, 64, 3, 1, 1, 1), 'conv_bf16x3_ws_kernel<3,2,1>', False),
    ((4, 32, 32, 1024, 3, 1, 1, 1), 'conv_bf16x3_ws_kernel<3,1,', False),
    ((4, 512, 512, 32, 3, 1, 1, 1), 'conv_bf16x3_kernel<1,4,2,1>|up1,down1', True),
    ((8, 512, 512, 32, 3, 1, 1, 1), 'conv_bf16x3_kernel<1,4,2,1>|up1,down1', False),      # 512 workgroups of 4-row tiles already: unsplit (round 5)
    ((4, 512, 512, 16, 3, 1, 1, 1), 'conv_bf16x3_kernel<1,4,2,1>|up1,down1', True),
    ((4, 512, 512, 8, 3, 1, 1, 1), 'conv_f32_small_kernel<3,1,2,1>|up1,down1', False),
    ((4, 513, 512, 4, 3, 1, 1, 1), 'conv_f32_small_kernel<3,2,1,1>|up1,down1', False),
    # stride 2 (after the Blur: 2H + 1 -> H); 17 -> 8 at B = 8 in two sample groups of the small-plane kernel (round 5: was conv_mfma_kernel)
    # (round 6: the layers with >= 192 eight-row tiles x samples x oc blocks on the wave-specialised E / O kernel; 128- or 64-channel output blocks)
    ((4, 128, 256, 257, 3, 1, 2, 0), 'conv_s2ws_bf16x3_kernel<2>|up1,down2', False),
    ((8, 32, 64, 1025, 3, 1, 2, 0), 'conv_s2ws_bf16x3_kernel<1>|up1,down2', False),
    ((4, 256, 512, 129, 3, 1, 2, 0), 'conv_s2ws_bf16x3_kernel<2>|up1,down2', False),
    ((8, 512, 512, 65, 3, 1, 2, 0), 'conv_bf16x3_kernel<1,4,2,1>|up1,down2', False),      # 128 workgroups of that form: one-role kernel
    ((4, 512, 512, 65, 3, 1, 2, 0), 'conv_bf16x3_kernel<1,4,2,1>|up1,down2', True),
    ((8, 512, 512, 17, 3, 1, 2, 0), 'conv_f32_small_kernel<3,2,1,2>|up1,down2', False),
    ((4, 512, 512, 9, 3, 1, 2, 0), 'conv_f32_small_kernel<3,2,1,2>|up1,down2', False),
    # transposed: small planes on the zero-stuffed small-plane kernel / split over K; H x W main region + edge kernel only with >= 512 main-region workgroups
    ((4, 512, 512, 4, 3, 2, 1, 2), 'conv_f32_small_kernel<3,1,2,1>|up2,down1', False),
    ((8, 512, 512, 4, 3, 2, 1, 2), 'conv_f32_small_kernel<3,1,2,1>|up2,down1', False),
    ((4, 512, 512, 8, 3, 2, 1, 2), 'convt_fused_bf16x3_kernel<2,2,2,16>|up2', True),
    ((4, 512, 512, 16, 3, 2, 1, 2), 'convt_fused_bf16x3_kernel<2,2,2,32>|up2', True),
    ((4, 512, 512, 32, 3, 2, 1, 2), 'convt_fused_bf16x3_kernel<2,2,2,16>|up2', False),
    ((8, 512, 512, 32, 3, 2, 1, 2), 'convt_fused_bf16x3_kernel<2,2,2,32>+edge|up2', False),
    ((2, 512, 256, 64, 3, 2, 1, 2), 'convt_fused_bf16x3_kernel<2,2,2,16>|up2', False),
    ((4, 512, 256, 64, 3, 2, 1, 2), 'convt_fused_bf16x3_kernel<2,2,2,32>+edge|up2', False),
    ((4, 256, 128, 128, 3, 2, 1, 2), 'convt_fused_bf16x3_kernel<2,2,2,32>+edge|up2', False),
    ((4, 128, 64, 256, 3, 2, 1, 2), 'convt_fused_bf16x3_kernel<2,2,2,16>|up2', False),      # < 256 input channels: bound by its stores, one-kernel form
    ((4, 64, 32, 512, 3, 2, 1, 2), 'convt_fused_bf16x3_kernel<1,4,2,32>|up2', False),
]


@pytest.mark.parametrize('shape,prefix,split', STEP_SHAPES)
def test_step_shapes_reach_their_kernels(shape, prefix, split):
    name, slice_bytes = _probe(*shape)
    assert name.startswith(prefix), (shape, name)
    assert (slice_bytes > 0) == split, (shape, name, slice_bytes)


def test_exact_fp32_mode_takes_the_same_small_plane_kernels():
    for shape in [(4, 512, 512, 8, 3, 1, 1, 1), (4, 512, 512, 4, 3, 2, 1, 2), (8, 512, 512, 17, 3, 1, 2, 0)]:
        assert _probe(*shape, mode='f32')[0] == _probe(*shape)[0], shape
    assert _probe(4, 512, 512, 64, 3, 1, 1, 1, mode='f32')[0].startswith('conv_mfma_kernel')


def test_split_workspace_covers_the_slices():
    """gc_conv2d_bf16x3_workspace >= packed weights + K slices for a split transposed layer; the slices are whole output tensors."""
    b, K, N, h = 4, 512, 512, 8
    o = _out(h, 3, 2, 1, 2)
    desc = _lib.ConvDesc(b, K, N, h, h, o, o, 3, 3, 2, 1, 2, 2)
    lib = _lib.load()
    slices = lib.gc_conv2d_bf16x3_splitk_bytes(desc)
    per_slice = b * N * o * o * 4
    assert slices > 0 and slices % per_slice == 0 and 2 <= slices // per_slice <= K // 64
    assert lib.gc_conv2d_bf16x3_workspace(desc) >= lib.gc_conv2d_bf16x3_packed_bytes(desc) + slices
