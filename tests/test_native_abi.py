"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every symbol include/rqamd.h
declares; the binding refuses to run without it (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


@pytest.fixture(scope='module')
def lib_path():
    if not os.path.exists(HIPCC):
        pytest.skip('hipcc not available')
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.build()
    from rqvae import _native
    return _native.LIB_PATH


def test_header_symbols_exported(lib_path):
    header = open(os.path.join(ROOT, 'include', 'rqamd.h')).read()
    declared = set(re.findall(r'\b(rqamd_[a-z0-9_]+)\s*\(', header))
    assert len(declared) >= 17
    lib = ctypes.CDLL(lib_path)
    for name in sorted(declared):
        assert hasattr(lib, name), f'{name} declared in include/rqamd.h but not exported'
    from rqvae import _native
    assert declared == set(_native.EXPORTS)
    lib.rqamd_abi_version.restype = ctypes.c_int
    assert lib.rqamd_abi_version() == _native.ABI_VERSION == 7
    # the fp16 build of the RQ-Transformer engine (sample(amp=True)): the rqamd_rqt_* subset of the same ABI
    assert os.path.exists(_native.LIB16_PATH)
    lib16 = ctypes.CDLL(_native.LIB16_PATH)
    for name in _native.EXPORTS_F16:
        assert hasattr(lib16, name), f'{name} missing from librqamd_f16.so'
    assert {n for n in declared if n.startswith('rqamd_rqt_')} <= set(_native.EXPORTS_F16)
    lib16.rqamd_abi_version.restype = ctypes.c_int
    assert lib16.rqamd_abi_version() == _native.ABI_VERSION


def test_status_codes_without_gpu(lib_path):
    """argument validation happens before any HIP call, so it is observable on a GPU-less host"""
    lib = ctypes.CDLL(lib_path)
    lib.rqamd_last_error.restype = ctypes.c_char_p
    assert lib.rqamd_rqt_create(None, None) == -1
    assert b'null' in lib.rqamd_last_error()
    from rqvae._native import RqtConfig
    cfg = RqtConfig(100, 3, 1, 1, 10, 64, 1, 1, 8, 8, 4, 0)          # embed_dim not a multiple of n_head (attentions.py:46 asserts)
    h = ctypes.c_void_p()
    assert lib.rqamd_rqt_create(ctypes.byref(cfg), ctypes.byref(h)) == -2
    assert b'head_dim' in lib.rqamd_last_error()
    cfg = RqtConfig(1024, 2, 1, 1, 10, 64, 1, 1, 8, 8, 4, 0)         # head_dim 512 > 256
    assert lib.rqamd_rqt_create(ctypes.byref(cfg), ctypes.byref(h)) == -2
    assert b'head_dim' in lib.rqamd_last_error()
    assert lib.rqamd_rqt_set_option(None, b'head.n_head', 2) == -1
    assert lib.rqamd_vae_decode(None, None, 1, None, None) == -1
    lib.rqamd_rq_quantize.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int64, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_void_p]
    assert lib.rqamd_rq_quantize(None, None, None, None, 4, 0, 256, None, None, None, 0, None) == 0     # empty input is a no-op


def test_no_cpu_fallback(lib_path, monkeypatch):
    from rqvae import _native
    with pytest.raises(_native.RqamdError, match='CPU'):
        _native.ptr(torch.zeros(4))
    monkeypatch.setattr(_native, '_lib', None)
    monkeypatch.setattr(_native, 'LIB_PATH', '/nonexistent/librqamd.so')
    with pytest.raises(_native.RqamdError, match='no CPU fallback'):
        _native.lib()


def test_binding_has_no_host_pointer_switch():
    """the emulator tests swap the library / pointer marshalling from OUTSIDE (tests/emu/emu_binding.py); the shipped
    binding itself has no flag or entry point that accepts host memory"""
    src = open(os.path.join(ROOT, 'rq-vae-transformer_amd', 'rqvae', '_native.py')).read()
    assert '_allow_host_pointers' not in src and '_load_for_testing' not in src


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'rq-vae-transformer_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f
