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


def test_every_switch_of_the_library_is_an_option_and_documented():
    """no switch of the library depends on the process environment alone, and none is undocumented.  Every name the
    native sources look up (opt_str(ctx, "JOXSZ_...")) is in the library's option table (kOptions: what jx_set_option accepts), every
    name of that table is looked up somewhere, and every one of them appears in the option table of include/joxsz_hip.h; nothing in
    csrc/ calls getenv except the one fall-back inside opt_str."""
    csrc = os.path.join(ROOT, 'joxsz_amd', 'csrc')
    used, text_all = set(), ''
    for f in os.listdir(csrc):
        if f.endswith(('.hip', '.hpp', '.cpp')):
            text = open(os.path.join(csrc, f)).read()
            text_all += text
            used |= set(re.findall(r'opt_str\(ctx, "(JOXSZ_[A-Z0-9_]+)"\)', text))
    assert len(re.findall(r'\bgetenv\(', text_all)) == 1 and 'env_str(' not in text_all
    table = re.search(r'kOptions\[\] = \{(.*?)\};', text_all, flags=re.S).group(1)
    table = set(re.findall(r'"(JOXSZ_[A-Z0-9_]+)"', table))
    assert used == table, (sorted(used - table), sorted(table - used))
    header = open(os.path.join(ROOT, 'include', 'joxsz_hip.h')).read()
    doc = header[header.index('One switch of the library'):header.index('int  jx_set_option')]
    missing = [n for n in sorted(table) if n not in doc]
    assert not missing, missing
    # the Python side: its own three variables are documented in the README
    readme = open(os.path.join(ROOT, 'README.md')).read()
    for n in ('JOXSZ_LIB', 'JOXSZ_QUIET', 'JOXSZ_ROUTE'):
        assert n in readme, n


def test_set_option_refuses_unknown_names(lib):
    """jx_set_option needs no device to refuse a name it does not know... but it needs a context: checked on the GPU (test_gpu_configs.py);
    here: the symbol is there and the binding passes options through."""
    import inspect
    assert 'options' in inspect.signature(hip_backend.HipContext.__init__).parameters
    assert hasattr(lib, 'jx_set_option') and hasattr(lib, 'jx_audit')
