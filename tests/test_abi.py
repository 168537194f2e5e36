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
    assert lib.tamgcn_version() == _lib.ABI_VERSION == 401
    # argument validation happens before any launch => usable on a CPU-only box
    assert lib.tamgcn_reduce_sum(None, 0, 0, 0, 1.0, 0, None, None) < 0
    assert b'tamgcn_reduce_sum' in lib.tamgcn_last_error()
    assert lib.tamgcn_ctrgc_lds_bytes(3, 20, 8) > 0
    assert lib.tamgcn_ctrgc_lds_bytes(3, 20, 8) <= 160 * 1024
    assert lib.tamgcn_ctrgc_lds_bytes(3, 25, 32) > 0
    assert 0 < lib.tamgcn_ctrgc_lds_bytes(3, 64, 32) <= 160 * 1024     # V = 64: the tiled family (x3 GEMM + MFMA aggregation)
    assert lib.tamgcn_ctrgc_lds_bytes(3, 48, 32) < 0        # no tiling for this V: rejected, not mis-run


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from tam_gcn_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.TamgcnLibraryError, match='no CPU fallback'):
        _lib.load()


def test_abi_rejects_bad_arguments_without_touching_the_gpu():
    """Argument checks run before any HIP call: they can be exercised on a machine without a GPU.  Every entry point
    returns a negative status and leaves a message in tamgcn_last_error(); unsupported geometries are refused, not mis-run."""
    import ctypes as C
    from tam_gcn_amd import build, _lib
    build.build()
    lib = C.CDLL(_lib.LIB_PATH)
    lib.tamgcn_last_error.restype = C.c_char_p
    for fn, args in [('tamgcn_conv', (None, None)), ('tamgcn_wgrad', (None, None)),
                     ('tamgcn_ctrgc_fwd', (None, None, None, None, None)), ('tamgcn_reduce_multi', (None, 0, None)),
                     ('tamgcn_ctrgc_build_e', (None, None, None))]:
        rc = getattr(lib, fn)(*args)
        assert rc < 0, fn
        assert fn.encode() in lib.tamgcn_last_error(), (fn, lib.tamgcn_last_error())
    assert lib.tamgcn_conv_nparts(None) == -1 and lib.tamgcn_wgrad_max_split(None) == -1
    assert lib.tamgcn_ctrgc_lds_bytes(3, 20, 8) > 64 * 1024          # N-UCLA tiles: LDS resident
    assert lib.tamgcn_ctrgc_lds_bytes(3, 25, 8) > 0                  # NTU
    assert lib.tamgcn_ctrgc_lds_bytes(3, 64, 32) > 0                 # V = 64: tiled family
    assert lib.tamgcn_ctrgc_tiled_supported(64) == 1 and lib.tamgcn_ctrgc_tiled_supported(25) == 2 and lib.tamgcn_ctrgc_tiled_supported(40) == 0
    assert lib.tamgcn_ctrgc_lds_bytes(3, 40, 8) == -1                # neither family
    assert lib.tamgcn_ctrgc_lds_bytes(2, 20, 8) == -1                # subsets: 1 or 3
    # a descriptor with impossible sizes is refused with its own message
    d = _lib.ConvDesc()
    d.N, d.K, d.M, d.T_in, d.T_out, d.V, d.KT = 0, 4, 4, 4, 4, 20, 1
    one = C.c_float(0.0)
    d.src.x1 = C.addressof(one); d.w = C.addressof(one); d.y = C.addressof(one)
    assert lib.tamgcn_conv(C.byref(d), None) < 0 and b'bad dims' in lib.tamgcn_last_error()
