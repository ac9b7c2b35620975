"""The C-ABI library loads and exports every symbol include/gancontrol_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def declared_symbols():
    text = open(os.path.join(REPO, 'include', 'gancontrol_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(gc_[a-z0-9_]+)\s*\(', text)))


def test_header_and_binding_table_agree():
    from gan_control_amd import _lib
    assert declared_symbols() == sorted(_lib.SIGNATURES.keys())


def test_library_exports_every_declared_symbol():
    from gan_control_amd import _lib
    path = _lib.library_path()
    if not os.path.exists(path):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(path)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert _lib.load().gc_abi_version() == _lib.ABI_VERSION


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before any launch, so these calls are safe on a GPU-less host."""
    from gan_control_amd import _lib
    lib = _lib.load()
    assert lib.gc_upfirdn2d_f32(None, None, None, 1, 4, 4, 4, 4, 4, 4, 1, 1, 1, 1, 0, 0, 1, None) == -1
    assert b'null' in lib.gc_last_error()
    d = _lib.ConvDesc(1, 4, 4, 8, 8, 8, 8, 5, 5, 1, 1, 2, 2)
    assert lib.gc_conv2d_f32(d, 1, 1, None, None, 1, None) == -2          # 5x5 taps: unsupported
    assert b'taps' in lib.gc_last_error()
    d = _lib.ConvDesc(1, 4, 4, 8, 8, 8, 8, 3, 3, 2, 2, 1, 1)
    assert lib.gc_conv2d_f32(d, 1, 1, None, None, 1, None) == -2          # up = down = 2
    assert lib.gc_conv2d_wgrad_workspace(_lib.ConvDesc(4, 512, 512, 64, 64, 64, 64, 3, 3, 1, 1, 1, 1)) > 0
    assert lib.gc_bias_act_f32(1, None, 1, None, 1, 1, 1, 1, 0.2, 1.0, None) == -1   # noise without weight


def test_missing_library_fails_loudly(monkeypatch):
    from gan_control_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setenv('GANCONTROL_HIP_LIB', '/nonexistent/libgancontrol_hip.so')
    with pytest.raises(RuntimeError, match='no CPU or PyTorch fallback'):
        _lib.load()


def test_struct_mirrors_match_the_header_and_the_library(tmp_path):
    """gc_conv_desc / gc_conv_epilogue grew fields under an unchanged GC_ABI_VERSION.  Three views of every struct must
    agree: the header compiled by the host C compiler, the ctypes mirrors, and the built library's gc_struct_sizes()."""
    import subprocess
    from gan_control_amd import _lib
    names = ['gc_conv_desc', 'gc_conv_epilogue', 'gc_wlayout_group', 'gc_wpack_group', 'gc_glin_group', 'gc_wsq_group']
    src = tmp_path / 'sizes.c'
    src.write_text('#include <stdio.h>\n#include "gancontrol_hip.h"\nint main(void) { printf("%d %d", GC_ABI_VERSION, GC_STRUCT_COUNT);'
                   + ''.join(' printf(" %%zu", sizeof(%s));' % n for n in names) + ' return 0; }\n')
    exe = tmp_path / 'sizes'
    subprocess.run(['gcc', '-I', os.path.join(REPO, 'include'), str(src), '-o', str(exe)], check=True)
    out = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    mirrors = [ctypes.sizeof(c) for c in _lib.STRUCTS]
    assert out[0] == _lib.ABI_VERSION and out[1] == len(names) == len(_lib.STRUCTS)
    assert out[2:] == mirrors
    lib = _lib.load()
    got = (ctypes.c_size_t * len(names))()
    assert lib.gc_struct_sizes(got, len(names)) == len(names)
    assert list(got) == mirrors


def test_stale_library_is_rejected(monkeypatch):
    """A library whose structs differ from the binding's mirrors (built from another header under the same version number) must not load."""
    from gan_control_amd import _lib
    _lib.load()

    class OldConvDesc(ctypes.Structure):           # gc_conv_desc before in_pitch / out_pitch
        _fields_ = _lib.ConvDesc._fields_[:-2]

    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'STRUCTS', (OldConvDesc,) + tuple(_lib.STRUCTS[1:]))
    with pytest.raises(RuntimeError, match='built from another header'):
        _lib.load()
