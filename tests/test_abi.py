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
