"""The C-ABI library loads and exports every symbol include/tamgcn.h declares
(no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'tamgcn.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(tamgcn_[a-z0-9_]+)\s*\(', src)))


def test_library_builds_and_exports_every_declared_symbol():
    from tam_gcn_amd import build, _lib
    build.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f'{n} declared in tamgcn.h but not exported'
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)


def test_version_and_error_text_callable_without_gpu():
    from tam_gcn_amd import _lib
    lib = _lib.load()
    assert lib.tamgcn_version() == 100
    # argument validation happens before any launch => usable on a CPU-only box
    assert lib.tamgcn_reduce_sum(None, 0, 0, 0, 1.0, 0, None, None) < 0
    assert b'tamgcn_reduce_sum' in lib.tamgcn_last_error()
    assert lib.tamgcn_ctrgc_lds_bytes(3, 20, 8) > 0
    assert lib.tamgcn_ctrgc_lds_bytes(3, 20, 8) <= 160 * 1024
    assert lib.tamgcn_ctrgc_lds_bytes(3, 25, 32) > 0
    assert lib.tamgcn_ctrgc_lds_bytes(3, 64, 32) < 0        # V=64 tile not built yet: rejected, not mis-run


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from tam_gcn_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.TamgcnLibraryError, match='no CPU fallback'):
        _lib.load()
