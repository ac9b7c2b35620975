"""ctypes binding of the C-ABI library declared in include/gancontrol_hip.h.

The product path has NO fallback: if ``libgancontrol_hip.so`` is missing or a kernel call
fails, a RuntimeError is raised.  PyTorch is only the memory/stream substrate here: tensors
provide device pointers, ``torch.cuda.current_stream`` provides the hipStream_t.
"""
import ctypes
import os

import torch  # noqa: F401  (must be imported first: it loads the HIP runtime the library binds to)

_HERE = os.path.dirname(os.path.abspath(__file__))
_DEFAULT = os.path.normpath(os.path.join(_HERE, '..', 'csrc', 'libgancontrol_hip.so'))

ABI_VERSION = 2

_c_float_p = ctypes.c_void_p
_i32, _i64, _f32, _vp, _sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t


class ConvDesc(ctypes.Structure):
    """Mirror of ``gc_conv_desc``."""
    _fields_ = [(n, _i32) for n in ('batch', 'in_ch', 'out_ch', 'in_h', 'in_w', 'out_h', 'out_w',
                                    'kh', 'kw', 'up', 'down', 'pad_y', 'pad_x', 'in_pitch', 'out_pitch')]


class ConvEpilogue(ctypes.Structure):
    """Mirror of ``gc_conv_epilogue``."""
    _fields_ = [('bias', _vp), ('noise', _vp), ('noise_w', _vp), ('slope', _f32), ('gain', _f32), ('activate', _i32), ('residual', _vp)]


class GlinGroup(ctypes.Structure):
    """Mirror of ``gc_glin_group``."""
    _fields_ = [('x', _vp), ('w', _vp), ('bias', _vp), ('y', _vp), ('n', _i32), ('k', _i32), ('x_stride', _i64), ('alpha', _f32), ('beta', _f32)]


class WLayoutGroup(ctypes.Structure):
    """Mirror of ``gc_wlayout_group``."""
    _fields_ = [('src', _vp), ('dst', _vp), ('taps', _i32), ('k', _i32), ('n', _i32), ('flip_taps', _i32), ('src_stride', _i64 * 3), ('dst_stride', _i64 * 3), ('scale', _f32)]


class WPackGroup(ctypes.Structure):
    """Mirror of ``gc_wpack_group``."""
    _fields_ = [('desc', ConvDesc), ('w', _vp), ('packed', _vp), ('packed_bytes', _sz)]


class WsqGroup(ctypes.Structure):
    """Mirror of ``gc_wsq_group``."""
    _fields_ = [('w', _vp), ('g', _vp), ('out', _vp), ('rows', _i32), ('taps', _i32)]


# the mirrors above in the header's declaration order (gc_struct_sizes)
STRUCTS = (ConvDesc, ConvEpilogue, WLayoutGroup, WPackGroup, GlinGroup, WsqGroup)

# name -> (restype, argtypes); kept in one table so tests can check every exported symbol
SIGNATURES = {
    'gc_abi_version': (_i32, []),
    'gc_last_error': (ctypes.c_char_p, []),
    'gc_struct_sizes': (_i32, [ctypes.POINTER(_sz), _i32]),
    'gc_source_hash': (ctypes.c_char_p, []),
    'gc_upfirdn2d_f32': (_i32, [_vp, _vp, _vp] + [_i32] * 14 + [_vp]),
    'gc_upfirdn2d_act_f32': (_i32, [_vp, _vp, _vp] + [_i32] * 11 + [_vp, _vp, _vp, _f32, _f32, _vp]),
    'gc_bias_act_f32': (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i64, _f32, _f32, _vp]),
    'gc_bias_act_bwd_f32': (_i32, [_vp, _vp, _vp, _i64, _f32, _f32, _vp]),
    'gc_bias_act_bwd_chunks': (_i32, [_i64]),
    'gc_bias_act_bwd_reduce_f32': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i64, _f32, _f32, _vp]),
    'gc_bias_act_bwd_reduce_self_f32': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i64, _f32, _f32, _vp]),
    'gc_bias_act_bwd_reduce_adjoint_f32': (_i32, [_vp] * 13 + [_i32, _i32, _i64, _f32, _f32, _vp]),
    'gc_plane_dot_f32': (_i32, [_vp, _vp, _vp, _i32, _i64, _vp]),
    'gc_rows_sum_div_f32': (_i32, [_vp, _vp, _vp, _i32, _i32, _vp]),
    'gc_conv2d_bn_relu_f32': (_i32, [_vp, _vp, _vp, _vp, _vp] + [_i32] * 13 + [_vp]),
    'gc_pool2d_f32': (_i32, [_vp, _vp] + [_i32] * 10 + [_vp]),
    'gc_global_avgpool_f32': (_i32, [_vp, _vp, _i32, _i32, _vp]),
    'gc_resize_bilinear_f32': (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _vp]),
    'gc_channel_sum_workspace': (_sz, [_i32, _i32, _i64]),
    'gc_channel_sum_f32': (_i32, [_vp, _vp, _i32, _i32, _i64, _vp, _sz, _vp]),
    'gc_conv2d_f32': (_i32, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp]),
    'gc_conv2d_fused_f32': (_i32, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, ctypes.POINTER(ConvEpilogue), _vp, _vp]),
    'gc_conv2d_f32_workspace': (_sz, [ctypes.POINTER(ConvDesc)]),
    'gc_small_gemm_ok': (_i32, [_i32, _i32, _i32, _i64, _i64]),
    'gc_small_gemm_f32': (_i32, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _f32, _f32, _vp, _i32, _i32, _i32, _vp]),
    'gc_conv2d_fused_f32_ws': (_i32, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, ctypes.POINTER(ConvEpilogue), _vp, _vp, _sz, _vp]),
    'gc_conv2d_fused_bf16x3_f32': (_i32, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, ctypes.POINTER(ConvEpilogue), _vp, _vp, _sz, _vp]),
    'gc_conv2d_bf16x3_workspace': (_sz, [ctypes.POINTER(ConvDesc)]),
    'gc_conv2d_bf16x3_packed_bytes': (_sz, [ctypes.POINTER(ConvDesc)]),
    'gc_conv2d_bf16x3_splitk_bytes': (_sz, [ctypes.POINTER(ConvDesc)]),
    'gc_conv2d_pack_weights_bf16x3': (_i32, [ctypes.POINTER(ConvDesc), _vp, _vp, _sz, _vp]),
    'gc_conv2d_fused_bf16x3_packed_f32': (_i32, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _sz, _vp, _vp, ctypes.POINTER(ConvEpilogue), _vp, _vp, _sz, _vp]),
    'gc_conv2d_fused_bf16_packed_f32': (_i32, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _sz, _vp, _vp, ctypes.POINTER(ConvEpilogue), _vp, _vp, _sz, _vp]),
    'gc_conv2d_wgrad_bf16_f32': (_i32, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'gc_conv2d_variant_name': (_i32, [ctypes.POINTER(ConvDesc), _i32, ctypes.c_char_p, _i32]),
    'gc_conv2d_out_pitch': (_i32, [ctypes.POINTER(ConvDesc), _i32]),
    'gc_conv2d_in_pitch_ok': (_i32, [ctypes.POINTER(ConvDesc), _i32, _i32]),
    'gc_upfirdn2d_actbwd_tiles': (_i32, [_i32, _i32]),
    'gc_upfirdn2d_actbwd_f32': (_i32, [_vp] * 7 + [_i32] * 12 + [_f32, _f32, _vp]),
    'gc_upfirdn2d_mask_f32': (_i32, [_vp, _vp, _vp] + [_i32] * 12 + [_vp, _f32, _f32, _vp]),
    'gc_upfirdn2d_pitched_f32': (_i32, [_vp, _vp, _vp] + [_i32] * 14 + [_vp, _vp, _vp, _f32, _f32, _vp]),
    'gc_plane_dot_pitched_chunks': (_i32, [_i32]),
    'gc_plane_dot_pitched_f32': (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    'gc_conv2d_bf16x3_f32': (_i32, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'gc_conv2d_wgrad_workspace': (_sz, [ctypes.POINTER(ConvDesc)]),
    'gc_conv2d_wgrad_f32': (_i32, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'gc_conv2d_wgrad_bf16x3_workspace': (_sz, [ctypes.POINTER(ConvDesc)]),
    'gc_conv2d_wgrad_bf16x3_f32': (_i32, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'gc_conv2d_wgrad_samples_workspace': (_sz, [ctypes.POINTER(ConvDesc), _i32]),
    'gc_conv2d_wgrad_samples_f32': (_i32, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'gc_conv2d_wgrad_samples_bf16x3_f32': (_i32, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'gc_conv2d_wgrad_samples_bf16_f32': (_i32, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'gc_wgrad_samples_contract_workspace': (_sz, [_i32, _i32, _i32]),
    'gc_wgrad_samples_contract_f32': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    'gc_affine_warp_bilinear_f32': (_i32, [_vp, _vp, _vp] + [_i32] * 7 + [_vp]),
    'gc_reflect_pad_f32': (_i32, [_vp, _vp] + [_i32] * 8 + [_vp]),
    'gc_pw_act_wgrad_workspace': (_sz, [_i32, _i32, _i32, _i64]),
    'gc_pw_act_wgrad_f32': (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _f32, _f32, _vp, _sz, _vp]),
    'gc_pw_act_dgrad_f32': (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _f32, _f32, _vp]),
    'gc_grouped_linear_f32': (_i32, [ctypes.POINTER(GlinGroup), _i32, _i32, _vp]),
    'gc_grouped_linear_bwd_x_f32': (_i32, [ctypes.POINTER(GlinGroup), _i32, _i32, _vp]),
    'gc_grouped_linear_bwd_w_f32': (_i32, [ctypes.POINTER(GlinGroup), _i32, _i32, _vp]),
    'gc_weight_sq_grouped_f32': (_i32, [ctypes.POINTER(WsqGroup), _i32, _vp]),
    'gc_weight_sq_bwd_grouped_f32': (_i32, [ctypes.POINTER(WsqGroup), _i32, _vp]),
    'gc_weight_layout_grouped_f32': (_i32, [ctypes.POINTER(WLayoutGroup), _i32, _vp]),
    'gc_conv2d_pack_weights_bf16x3_grouped': (_i32, [ctypes.POINTER(WPackGroup), _i32, _vp]),
    'gc_weight_layout_f32': (_i32, [_vp, _vp, _i32, _i32, _i32, ctypes.POINTER(_i64 * 3), ctypes.POINTER(_i64 * 3), _i32, _f32, _vp]),
}

_lib = None


def source_hash():
    """The hash the LOADED library was stamped with at build time (gc_source_hash: sha256 over every kernel source, header and the compiler
    flags; alt builds of tools/build_alt.sh carry their extra flags in it): what a counter file collected on one build is tagged with, so that it
    is never quoted for another (bench.py roofline.traffic / mfma_busy, tools/pmc_mix.py)."""
    v = load().gc_source_hash()
    return v.decode() if v else None


def library_path():
    return os.environ.get('GANCONTROL_HIP_LIB', _DEFAULT)


def load():
    """Load (once) and return the ctypes handle; raise RuntimeError if the library is unusable."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f'gan_control_amd: HIP kernel library not found at {path}. Build it with '
            f'`make -C gan-control_amd/csrc` (or `python -c "import __graft_entry__ as g; g.build()"`). '
            f'There is no CPU or PyTorch fallback for the hot path.')
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:
        raise RuntimeError(f'gan_control_amd: cannot load {path}: {e}') from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f'gan_control_amd: {path} does not export {name}') from e
        fn.restype, fn.argtypes = res, args
    if lib.gc_abi_version() != ABI_VERSION:
        raise RuntimeError(f'gan_control_amd: ABI version {lib.gc_abi_version()} != {ABI_VERSION}; rebuild the library')
    # a library built from an older header with the same version number would mis-read every descriptor: compare struct sizes too
    theirs = (_sz * len(STRUCTS))()
    count = lib.gc_struct_sizes(theirs, len(STRUCTS))
    ours = [ctypes.sizeof(c) for c in STRUCTS]
    if count != len(STRUCTS) or list(theirs) != ours:
        raise RuntimeError(f'gan_control_amd: {path} was built from another header: struct sizes {list(theirs)[:count]} (library) != {ours} '
                           f'(binding: {[c.__name__ for c in STRUCTS]}); rebuild the library')
    _lib = lib
    return lib


class UnsupportedError(RuntimeError):
    """GC_ERR_UNSUPPORTED: a legal request outside what the kernels implement -- the one failure a caller may answer with another route."""


GC_ERR_UNSUPPORTED = -2        # include/gancontrol_hip.h


def check(rc, what):
    if rc != 0:
        msg = load().gc_last_error()
        kind = UnsupportedError if rc == GC_ERR_UNSUPPORTED else RuntimeError
        raise kind(f'{what} failed (code {rc}): {msg.decode() if msg else "?"}')


def ptr(t):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream_of(t):
    """The raw handle of the CURRENT stream of t's device (what the kernels launch on).  torch._C._cuda_getCurrentRawStream is the same query
    torch.cuda.current_stream(device).cuda_stream makes, without building a Stream object per call (~5 us, ~580 calls per training iteration)."""
    if _raw_stream is not None:
        idx = t.device.index
        return _raw_stream(idx if idx is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(t.device).cuda_stream


def row_pitch(t):
    """The row pitch (elements) of a 4-D tensor whose planes are ``H x pitch`` blocks laid out back to back with only the first ``W``
    columns of a row in use -- what the fused transposed convolution writes (gc_conv_desc.out_pitch) -- or 0 for anything else
    (contiguous tensors included: they need no special handling)."""
    if t is None or t.dim() != 4 or t.is_contiguous():
        return 0
    b, c, h, w = t.shape
    sb, sc, sh, sw = t.stride()
    # the kernels read such a tensor in place only when it is what they write themselves: rows a multiple of 32 floats apart, 16-byte
    # aligned -- an arbitrary column slice x[..., a:b] of a user tensor has the same stride pattern and must be copied instead
    if sw == 1 and sh > w and sc == h * sh and sb == c * sc and sh % 32 == 0 and t.data_ptr() % 16 == 0:
        return sh
    return 0


def require_cuda_f32(*tensors, pitched=()):
    """The kernels take contiguous float32 device memory; anything else is a caller error."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('gan_control_amd: the HIP hot path needs tensors on a GPU (got %s); there is no CPU fallback' % t.device)
        if t.dtype != torch.float32:
            raise RuntimeError('gan_control_amd: float32 tensors expected, got %s' % t.dtype)
        if not t.is_contiguous() and not (any(t is q for q in pitched) and row_pitch(t)):
            raise RuntimeError('gan_control_amd: internal error: non-contiguous tensor reached the kernel boundary')
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError('gan_control_amd: tensors on different devices (%s vs %s)' % (dev, t.device))
    return dev
