"""The C-ABI library loads and exports every function include/joxsz_hip.h declares
(no compute calls: this runs without a GPU), and the host mirror fails loudly
instead of falling back when no device can be used."""
import ctypes
import os
import re

import numpy as np
import pytest

from joxsz_amd import hip_backend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'joxsz_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(jx_[a-z0-9_]+)\s*\(', src)))


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(hip_backend.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return hip_backend.load_library()


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(hip_backend.EXPORTS) == names


def test_config_struct_layout():
    # 22 int32 then 7 doubles, no implicit padding (the header keeps the doubles 8-byte aligned)
    assert ctypes.sizeof(hip_backend.JxConfig) == 22 * 4 + 7 * 8
    assert hip_backend.JxConfig.step.offset == 88
    assert ctypes.sizeof(hip_backend.JxTiming) == 6 * 8 + 2 * 8 + 8


def test_strerror_and_bad_config(lib):
    assert lib.jx_strerror(0) == b'ok'
    assert b'tensor' in lib.jx_strerror(-3)
    cfg = hip_backend.JxConfig()                      # abi_version 0: refused before any device is touched
    h = ctypes.c_void_p()
    assert lib.jx_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert not h.value


def test_no_cpu_fallback_without_device(lib):
    """On a box without a GPU the product path must raise, never compute on the CPU."""
    if lib.jx_device_count() > 0:
        pytest.skip('a GPU is present')
    from joxsz_amd import datasets
    from joxsz_amd.posterior import JoxszPosterior
    pb = datasets.synthetic_problem(S=31, N=40, step=6., fwhm=8.5)
    with pytest.raises(hip_backend.JoxszHipError):
        JoxszPosterior(pb)


def test_missing_library_raises(tmp_path):
    with pytest.raises(hip_backend.JoxszHipError):
        hip_backend.load_library(str(tmp_path / 'nope.so'))


def test_product_does_not_import_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, 'joxsz_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hpp', '.hip', '.cpp', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', text, flags=re.M), f
                assert 'oracle/' not in text or f.endswith(('.hpp', '.cpp')), f
