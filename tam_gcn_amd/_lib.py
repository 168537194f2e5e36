"""ctypes binding of libtamgcn.so (the C ABI declared in include/tamgcn.h).

The product path has no CPU fallback: if the shared object is missing or does
not export the ABI, importing the op layer raises (fail loudly).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libtamgcn.so')

c_float_p = C.c_void_p          # device pointers travel as integers


class Src(C.Structure):
    _fields_ = [('x1', C.c_void_p), ('x2', C.c_void_p), ('coef', C.c_void_p),
                ('ctot', C.c_int), ('coff', C.c_int), ('act', C.c_int)]


class ConvDesc(C.Structure):
    _fields_ = [('src', Src),
                ('N', C.c_int), ('K', C.c_int), ('T_in', C.c_int), ('V', C.c_int),
                ('w', C.c_void_p), ('bias', C.c_void_p),
                ('M', C.c_int), ('KT', C.c_int), ('dil', C.c_int), ('stride', C.c_int), ('pad', C.c_int),
                ('wmode', C.c_int), ('up', C.c_int),
                ('y', C.c_void_p), ('yctot', C.c_int), ('ycoff', C.c_int),
                ('T_out', C.c_int), ('T_y', C.c_int), ('ostride', C.c_int),
                ('add1', C.c_void_p), ('add2', C.c_void_p), ('bcast', C.c_void_p), ('bcast_scale', C.c_float),
                ('mask', C.POINTER(Src)),
                ('aux', C.c_void_p), ('aux_center', C.c_void_p), ('auxctot', C.c_int), ('auxcoff', C.c_int),
                ('stats_part', C.c_void_p), ('stats_ctot', C.c_int), ('stats_coff', C.c_int),
                ('post_coef', C.c_void_p), ('post_ctot', C.c_int), ('post_act', C.c_int)]


TCONV_MAXB = 6


class TconvDesc(C.Structure):
    _fields_ = [('src', Src),
                ('N', C.c_int), ('T_in', C.c_int), ('T_out', C.c_int), ('V', C.c_int), ('Cb', C.c_int), ('nb', C.c_int),
                ('KT', C.c_int), ('stride', C.c_int), ('pool', C.c_int),
                ('dil', C.c_int * TCONV_MAXB), ('w', C.c_void_p * TCONV_MAXB), ('bias', C.c_void_p * TCONV_MAXB),
                ('y', C.c_void_p), ('yctot', C.c_int), ('ycoff', C.c_int),
                ('stats_part', C.c_void_p), ('stats_ctot', C.c_int),
                ('mask', C.POINTER(Src)), ('center', C.c_void_p)]


class WgradDesc(C.Structure):
    _fields_ = [('gy', Src), ('src', Src),
                ('N', C.c_int), ('M', C.c_int), ('K', C.c_int), ('T_in', C.c_int), ('T_out', C.c_int),
                ('V', C.c_int), ('KT', C.c_int), ('dil', C.c_int), ('stride', C.c_int), ('pad', C.c_int),
                ('part', C.c_void_p), ('nsplit', C.c_int)]


class ReduceDesc(C.Structure):
    _fields_ = [('part', C.c_void_p), ('out', C.c_void_p), ('nsplit', C.c_int), ('accumulate', C.c_int),
                ('stride_s', C.c_longlong), ('count', C.c_longlong), ('scale', C.c_float)]


class BnFwdDesc(C.Structure):
    _fields_ = [('part', C.c_void_p), ('part_ctot', C.c_int), ('part_coff', C.c_int), ('nparts', C.c_int), ('count', C.c_double),
                ('gamma', C.c_void_p), ('beta', C.c_void_p), ('running_mean', C.c_void_p), ('running_var', C.c_void_p),
                ('num_batches_tracked', C.c_void_p), ('momentum', C.c_float), ('eps', C.c_float), ('training', C.c_int),
                ('coef', C.c_void_p), ('save', C.c_void_p), ('coef_ctot', C.c_int), ('coef_coff', C.c_int), ('C', C.c_int)]


class BnBwdDesc(C.Structure):
    _fields_ = [('part', C.c_void_p), ('part_ctot', C.c_int), ('part_coff', C.c_int), ('nparts', C.c_int), ('count', C.c_double),
                ('gamma', C.c_void_p), ('save', C.c_void_p), ('save_ctot', C.c_int), ('save_coff', C.c_int), ('training', C.c_int),
                ('dgamma', C.c_void_p), ('dbeta', C.c_void_p), ('dbias_conv', C.c_void_p), ('coef', C.c_void_p),
                ('coef_ctot', C.c_int), ('coef_coff', C.c_int), ('C', C.c_int)]


class CtrgcDesc(C.Structure):
    _fields_ = [('N', C.c_int), ('Cin', C.c_int), ('Cout', C.c_int), ('S', C.c_int), ('R', C.c_int),
                ('T', C.c_int), ('V', C.c_int),
                ('x', Src),
                ('pq', C.c_void_p), ('w3', C.c_void_p), ('b3', C.c_void_p), ('w4', C.c_void_p),
                ('b4', C.c_void_p), ('A', C.c_void_p), ('alpha', C.c_void_p), ('E', C.c_void_p)]


class F2GcnDesc(C.Structure):
    _fields_ = [(k, C.c_int) for k in ('N', 'Cin', 'Cout', 'T', 'V', 'S', 'R', 'res_mode')] + \
               [(k, C.c_void_p) for k in ('x', 'w12', 'b12', 'w4', 'b4', 'A', 'alpha', 'w3', 'b3', 'sy', 'ty', 'wd', 'bd',
                                          'E', 'sum', 'diff', 'xpart')]


class F2GemmDesc(C.Structure):
    _fields_ = [(k, C.c_int) for k in ('N', 'K', 'M', 'T', 'V', 'mode', 'relu_rows')] + \
               [(k, C.c_void_p) for k in ('x', 'w', 'b', 'add', 'out')]


class F2TcnDesc(C.Structure):
    _fields_ = [(k, C.c_int) for k in ('N', 'Cin', 'Cout', 'T', 'V', 'stride', 'Cb', 'nb', 'ks', 'res_mode')] + \
               [('dil', C.c_int * 4), ('h', C.c_void_p), ('wt', C.c_void_p * 4), ('bt', C.c_void_p * 4)] + \
               [(k, C.c_void_p) for k in ('sp', 'tp', 'x', 'wr', 'br', 'out', 'xpart')]


# name -> (restype, argtypes); must list every symbol of include/tamgcn.h
_i, _p, _d, _f, _ll = C.c_int, C.c_void_p, C.c_double, C.c_float, C.c_longlong
_SP = C.POINTER(Src)
SIGNATURES = {
    'tamgcn_version': (_i, []),
    'tamgcn_last_error': (C.c_char_p, []),
    'tamgcn_last_kernel': (C.c_char_p, []),
    'tamgcn_ctrgc_lds_bytes': (_i, [_i, _i, _i]),
    'tamgcn_get_split_mode': (_i, []),
    'tamgcn_set_split_mode': (_i, [_i]),
    'tamgcn_conv_nparts': (_i, [C.POINTER(ConvDesc)]),
    'tamgcn_conv': (_i, [C.POINTER(ConvDesc), _p]),
    'tamgcn_wgrad_max_split': (_i, [C.POINTER(WgradDesc)]),
    'tamgcn_wgrad': (_i, [C.POINTER(WgradDesc), _p]),
    'tamgcn_stem_stats': (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p]),
    'tamgcn_stem_apply': (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p]),
    'tamgcn_head_pool_fwd': (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    'tamgcn_head_pool_bwd': (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    'tamgcn_head_fc_fwd': (_i, [_p, _p, _p, _i, _i, _i, _p, _p]),
    'tamgcn_head_fc_bwd': (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _p, _p]),
    'tamgcn_reduce_multi': (_i, [C.POINTER(ReduceDesc), _i, _p]),
    'tamgcn_reduce_sum': (_i, [_p, _i, _ll, _ll, _f, _i, _p, _p]),
    'tamgcn_bn_fwd_finalize': (_i, [_p, _i, _i, _i, _d, _p, _p, _p, _p, _p, _f, _f, _i, _p, _p, _i, _i, _i, _p]),
    'tamgcn_bn_bwd_finalize': (_i, [_p, _i, _i, _i, _d, _p, _p, _i, _i, _i, _p, _p, _p, _p, _i, _i, _i, _p]),
    'tamgcn_bn_fwd_finalize_multi': (_i, [C.POINTER(BnFwdDesc), _i, _p]),
    'tamgcn_coef_diff': (_i, [_p, _p, _p, _i, _i, _p]),
    'tamgcn_bn_bwd_finalize_multi': (_i, [C.POINTER(BnBwdDesc), _i, _p]),
    'tamgcn_tmean': (_i, [_SP, _i, _i, _i, _i, _p, _p]),
    'tamgcn_ctrgc_build_e': (_i, [C.POINTER(CtrgcDesc), _p, _p]),
    'tamgcn_ctrgc_fwd': (_i, [C.POINTER(CtrgcDesc), _p, _p, _p, _p]),
    'tamgcn_ctrgc_bwd_dx3': (_i, [C.POINTER(CtrgcDesc), _SP, _p, _p, _p]),
    'tamgcn_ctrgc_bwd_de_acc': (_i, [C.POINTER(CtrgcDesc), _SP, _p, _p, _p]),
    'tamgcn_ctrgc_bwd_de_tail': (_i, [C.POINTER(CtrgcDesc), _p, _p, _p, _p, _p, _p, _i, _p]),
    'tamgcn_ctrgc_tiled_supported': (_i, [_i]),
    'tamgcn_ctrgc_tiled_chunks': (_i, [_i]),
    'tamgcn_ctrgc_tiled_build_e': (_i, [C.POINTER(CtrgcDesc), _p, _p]),
    'tamgcn_ctrgc_tiled_agg_fwd': (_i, [C.POINTER(CtrgcDesc), _p, _p, _p, _p, _p]),
    'tamgcn_ctrgc_tiled_agg_bwd': (_i, [C.POINTER(CtrgcDesc), _SP, _p, _p, _p, _p]),
    'tamgcn_ctrgc_tiled_de_acc': (_i, [C.POINTER(CtrgcDesc), _SP, _p, _p, _p]),
    'tamgcn_ctrgc_tiled_de_tail': (_i, [C.POINTER(CtrgcDesc), _p, _p, _p, _p, _p, _p, _p]),
    'tamgcn_ew_nparts': (_i, [_i, _i, _i, _i]),
    'tamgcn_gcn_tail_fwd': (_i, [_SP, _SP, _SP, _i, _i, _i, _i, _p, _p]),
    'tamgcn_gcn_tail_bwd': (_i, [_p, _p, _SP, _p, _i, _i, _i, _i, _p, _p, _p, _p]),
    'tamgcn_gcn_mid_bwd': (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p]),
    'tamgcn_tconv_supported': (_i, [_i, _i, _i, _i, C.POINTER(C.c_int), _i, _i]),
    'tamgcn_tconv_nparts': (_i, [C.POINTER(TconvDesc), _i]),
    'tamgcn_tconv_fwd': (_i, [C.POINTER(TconvDesc), _p]),
    'tamgcn_tconv_bwd': (_i, [C.POINTER(TconvDesc), _p]),
    'tamgcn_tconv_wgrad_max_split': (_i, [C.POINTER(TconvDesc)]),
    'tamgcn_tconv_wgrad': (_i, [C.POINTER(TconvDesc), _p]),
    'tamgcn_maxpool_fwd': (_i, [_SP, _i, _i, _i, _i, _i, _p, _i, _i, _i, _p, _p]),
    'tamgcn_maxpool_post_fwd': (_i, [_SP, _i, _i, _i, _i, _i, _p, _i, _i, _i, _p, _p, _i, _p]),
    'tamgcn_maxpool_bwd': (_i, [_SP, _SP, _p, _i, _i, _i, _i, _i, _i, _p, _i, _i, _p, _p]),
    'tamgcn_add_act_fwd': (_i, [_SP, _SP, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    'tamgcn_add_act_bwd': (_i, [_p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p]),
    'tamgcn_apply': (_i, [_SP, _i, _i, _i, _i, _p, _i, _i, _p]),
    'tamgcn_score_fuse': (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p]),
    'tamgcn_ce_fwd': (_i, [_p, _p, _i, _i, _p, _p, _p]),
    'tamgcn_ce_bwd': (_i, [_p, _p, _i, _i, _p, _p]),
    'tamgcn_stream_derive': (_i, [_p, _i, _i, _i, _i, _i, _p, _i, _p, _p]),
    'tamgcn_feeder_transform': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p]),
    'tamgcn_f2_e': (_i, [C.POINTER(F2GcnDesc), _p]),
    'tamgcn_f2_gcn': (_i, [C.POINTER(F2GcnDesc), _p]),
    'tamgcn_f2_gemm': (_i, [C.POINTER(F2GemmDesc), _p]),
    'tamgcn_f2_tcn': (_i, [C.POINTER(F2TcnDesc), _p]),
}


class TamgcnLibraryError(RuntimeError):
    pass


_lib = None
ABI_VERSION = 401          # include/tamgcn.h TAMGCN_VERSION this binding's structs and signatures were written for


def load():
    """Load libtamgcn.so and bind every ABI symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so (SONAME libamdhip64.so.7) and finds it by file name
    # through an RPATH; libtamgcn.so asks for libamdhip64.so.7.  If we were loaded first the
    # process would end up with TWO HIP runtimes (ours from /opt/rocm, torch's bundled one) and
    # our launches fail with "no ROCm-capable device".  Loading torch first makes the dynamic
    # loader satisfy our DT_NEEDED with torch's already-loaded runtime (same SONAME).
    import torch  # noqa: F401
    path = os.environ.get('TAMGCN_LIB', LIB_PATH)      # instrumented side builds (tools/), same ABI
    if not os.path.exists(path):
        raise TamgcnLibraryError(
            f'{path} not found: the HIP extension is not built.  Run '
            '`python -m tam_gcn_amd.build` (needs hipcc, gfx950).  There is no CPU fallback.')
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise TamgcnLibraryError(f'{path} does not export {name}') from e
        fn.restype = res
        fn.argtypes = args
    got = lib.tamgcn_version()
    if got != ABI_VERSION:                             # a stale or side-built library with the same symbol set would be
        raise TamgcnLibraryError(                      # called with the wrong struct layouts and corrupt device memory
            f'{path} reports ABI version {got}, this binding is written for {ABI_VERSION}: rebuild with '
            '`python -m tam_gcn_amd.build --force`')
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().tamgcn_last_error()
        raise RuntimeError(f'{what} failed ({rc}): {msg.decode(errors="replace") if msg else "?"}')
