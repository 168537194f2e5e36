"""Thin Python plumbing over the C ABI: device pointers out of torch tensors,
output allocation through torch's caching allocator, the current HIP stream.
No arithmetic happens here."""
import ctypes as C
import os
import threading

import torch

from . import _lib
from ._lib import Src, ConvDesc, WgradDesc, CtrgcDesc, ReduceDesc, TconvDesc, TCONV_MAXB

RELU = 1


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    if t is None:
        return None
    if not (t.is_cuda and t.is_contiguous()):
        raise RuntimeError('tam_gcn_amd: expected a contiguous HIP (cuda) tensor; there is no CPU path')
    if t.dtype not in (torch.float32, torch.int64, torch.int32, torch.float64):
        raise RuntimeError(f'tam_gcn_amd: unsupported dtype {t.dtype}')
    return t.data_ptr()


class S:
    """A fused-prologue operand: value = act(c1*x1 + c2*x2 + c0) over (N, ctot, T, V)."""
    __slots__ = ('x1', 'x2', 'coef', 'coff', 'act', 'ctot')

    def __init__(self, x1, x2=None, coef=None, coff=0, act=0):
        self.x1, self.x2, self.coef, self.coff, self.act = x1, x2, coef, coff, act
        self.ctot = x1.shape[1]
        if x2 is not None and x2.shape != x1.shape:
            raise RuntimeError('tam_gcn_amd: x1/x2 geometry mismatch')
        if coef is not None and tuple(coef.shape) != (3, self.ctot):
            raise RuntimeError(f'tam_gcn_amd: coef shape {tuple(coef.shape)} != (3,{self.ctot})')

    def c(self):
        return Src(_ptr(self.x1), _ptr(self.x2), _ptr(self.coef), self.ctot, self.coff, self.act)


def _lib_():
    return _lib.load()


# Kernels index one launch's tensors with 32-bit element offsets inside a sample block and refuse tensors of >= 2^32
# elements; the V = 64, T = 512, C = 256 configuration at 256 clips has a 6.4e9-element x3.  Such calls are split over
# the clip dimension N (clips are independent; every tensor is contiguous in n) into launches below this bound.
CHUNK_ELEMS = int(os.environ.get("TAMGCN_CHUNK_ELEMS", str(2 ** 32 - 1)))


def n_chunks(N, per_clip):
    """[(n0, n1)] covering range(N) such that (n1 - n0) * per_clip <= CHUNK_ELEMS (at least one clip per chunk)."""
    c = max(1, CHUNK_ELEMS // max(1, per_clip))
    return [(i, min(N, i + c)) for i in range(0, N, c)]


def _slice_src(src, n0, n1):
    return S(src.x1[n0:n1], None if src.x2 is None else src.x2[n0:n1], src.coef, src.coff, src.act)


SLACK = 4          # floats readable behind every activation the ops allocate (see with_slack)


def empty(*shape, like):
    """fp32 tensor on like's device with SLACK floats of readable memory behind its last element."""
    n = 1
    for d_ in shape:
        n *= d_
    return torch.empty(n + SLACK, device=like.device, dtype=torch.float32)[:n].view(shape)


def empty_like(t):
    return empty(*t.shape, like=t)


def zeros_like(t):
    return empty(*t.shape, like=t).zero_()


def with_slack(t):
    """The 16-byte kernels may read up to 12 bytes past the last element of an activation whose rows (V joints) are not
    a multiple of 4 floats (the last 16-byte piece of the last frame).  Tensors from empty() carry that slack; a tensor
    that ends exactly at the end of its storage (a caller's input, an autograd-made gradient) is copied once."""
    if t is None or t.shape[-1] % 4 == 0:
        return t
    st = t.untyped_storage()
    if st.nbytes() - (t.storage_offset() + t.numel()) * t.element_size() >= 4 * SLACK:
        return t
    out = empty(*t.shape, like=t)
    out.copy_(t)
    return out


# ---------------------------------------------------------------------------
def conv(src, K, w, bias, M, KT=1, dil=1, stride=1, pad=0, wmode=0, up=1,
         y=None, ycoff=0, T_out=None, T_y=None, ostride=1, add1=None, add2=None,
         bcast=None, bcast_scale=0.0, mask=None, aux=None, aux_center=None, auxcoff=0, stats=False,
         post_coef=None, post_act=0):
    """y (N, yctot, T_y, V); returns (y, stats_part [2][yctot][nparts] or None).
    post_coef [3][yctot] / post_act: eval-mode fusion, y = act(c1*(conv + bias) + c0 + adds)."""
    if src.x1.shape[-1] % 4:
        src = S(with_slack(src.x1), with_slack(src.x2), src.coef, src.coff, src.act)
    x = src.x1
    N, _, T_in, V = x.shape
    if T_out is None:
        T_out = (T_in + 2 * pad - dil * (KT - 1) - 1) // stride + 1
    if y is None:
        if T_y is None:
            T_y = T_out
        y = empty(N, M, T_y, V, like=x)
    else:
        T_y = y.shape[2]
    per_clip = max(x.shape[1] * T_in * V, y.shape[1] * T_y * V)
    if N > 1 and N * per_clip > CHUNK_ELEMS:
        if stats or mask is not None or aux is not None:
            raise RuntimeError('tam_gcn_amd: a convolution over >= 2^31 elements cannot carry BatchNorm moments / a mask')
        for n0, n1 in n_chunks(N, per_clip):
            conv(_slice_src(src, n0, n1), K, w, bias, M, KT, dil, stride, pad, wmode, up, y=y[n0:n1], ycoff=ycoff, T_out=T_out,
                 ostride=ostride, add1=None if add1 is None else add1[n0:n1], add2=None if add2 is None else add2[n0:n1],
                 bcast=None if bcast is None else bcast[:, n0:n1].contiguous(), bcast_scale=bcast_scale,
                 post_coef=post_coef, post_act=post_act)
        return y, None
    d = ConvDesc()
    d.src = src.c()
    d.N, d.K, d.T_in, d.V = N, K, T_in, V
    d.w, d.bias = _ptr(w), _ptr(bias)
    d.M, d.KT, d.dil, d.stride, d.pad = M, KT, dil, stride, pad
    d.wmode, d.up = wmode, up
    d.y, d.yctot, d.ycoff = _ptr(y), y.shape[1], ycoff
    d.T_out, d.T_y, d.ostride = T_out, T_y, ostride
    d.add1, d.add2, d.bcast, d.bcast_scale = _ptr(add1), _ptr(add2), _ptr(bcast), bcast_scale
    mc = None
    if mask is not None:
        mc = mask.c()
        d.mask = C.pointer(mc)
    if aux is not None:
        d.aux, d.aux_center, d.auxctot, d.auxcoff = _ptr(aux), _ptr(aux_center), aux.shape[1], auxcoff
    if post_coef is not None:
        d.post_coef, d.post_ctot = _ptr(post_coef), post_coef.shape[1]
    d.post_act = int(post_act)
    part = None
    lib = _lib_()
    if stats:
        nparts = lib.tamgcn_conv_nparts(C.byref(d))
        if nparts <= 0:
            raise RuntimeError('tamgcn_conv: no tiling for this shape')
        part = empty(2, y.shape[1], nparts, like=x)
        d.stats_part, d.stats_ctot, d.stats_coff = _ptr(part), y.shape[1], ycoff
    _lib.check(lib.tamgcn_conv(C.byref(d), _stream()), 'tamgcn_conv')
    return y, part


# ---------------------------------------------------------------------------
# the MS-TCN second stage in one launch per direction (csrc/tconv.hip)
TCONV = os.environ.get('TAMGCN_TCONV', '1') != '0'         # 0: every branch through tamgcn_conv / tamgcn_maxpool_fwd (A/B, tests)


def tconv_supported(V, Cb, ks, dils, stride, T_in):
    """True when tamgcn_tconv_fwd / _bwd are built for this branch geometry (one kernel size for every branch)."""
    if not TCONV or not dils or len(dils) > TCONV_MAXB or any(k != ks[0] for k in ks):
        return False
    arr = (C.c_int * len(dils))(*[int(d_) for d_ in dils])
    return bool(_lib_().tamgcn_tconv_supported(V, Cb, int(ks[0]), len(dils), arr, stride, T_in))


def _tconv_desc(src, Cb, KT, dils, stride, ws, T_in, T_out, y, ycoff):
    d = TconvDesc()
    d.src = src.c()
    d.N, d.T_in, d.T_out, d.V, d.Cb, d.nb, d.KT, d.stride = src.x1.shape[0], T_in, T_out, src.x1.shape[3], Cb, len(dils), KT, stride
    for b, (dl, w) in enumerate(zip(dils, ws)):
        if tuple(w.shape[:3]) != (Cb, Cb, KT):
            raise RuntimeError(f'tamgcn_tconv: branch {b} weight {tuple(w.shape)}, expected ({Cb}, {Cb}, {KT}, 1)')
        d.dil[b], d.w[b] = int(dl), _ptr(w)
    d.y, d.yctot, d.ycoff = _ptr(y), y.shape[1], ycoff
    return d


def tconv_fwd(src, Cb, KT, dils, stride, ws, biases, pool, y, ycoff=0, stats=False):
    """Every temporal branch (and the pooled one) of an MS-TCN block: y[:, ycoff + b*Cb ...] for b < len(dils) (+ 1 with pool).
    src: S over (N, ctot, T_in, V) with the BatchNorm + ReLU prologue; returns the moment partials [2][yctot][nparts] or None."""
    N, _, T_in, V = src.x1.shape
    T_out = y.shape[2]
    if src.x1.shape[-1] % 4:
        src = S(with_slack(src.x1), None, src.coef, src.coff, src.act)
    d = _tconv_desc(src, Cb, KT, dils, stride, ws, T_in, T_out, y, ycoff)
    d.pool = int(bool(pool))
    for b, bi in enumerate(biases):
        d.bias[b] = _ptr(bi)
    lib = _lib_()
    part = None
    if stats:
        nparts = lib.tamgcn_tconv_nparts(C.byref(d), 0)
        if nparts <= 0:
            raise RuntimeError('tamgcn_tconv_fwd: no tiling for this shape')
        part = empty(2, y.shape[1], nparts, like=y)
        d.stats_part, d.stats_ctot = _ptr(part), y.shape[1]
    _lib.check(lib.tamgcn_tconv_fwd(C.byref(d), _stream()), 'tamgcn_tconv_fwd')
    return part


def tconv_bwd(gy, Cb, KT, dils, stride, ws, mask, center, dh, dcoff=0):
    """Data gradient of the temporal branches into dh[:, dcoff + b*Cb ...] (N, dctot, T_in, V), masked by relu(mask) > 0;
    returns the entry BatchNorm's backward moment partials [2][dctot][nparts]."""
    T_in, T_out = dh.shape[2], gy.x1.shape[2]
    if gy.x1.shape[-1] % 4:
        gy = S(with_slack(gy.x1), with_slack(gy.x2), gy.coef, gy.coff, gy.act)
        mask = S(with_slack(mask.x1), None, mask.coef, mask.coff, mask.act)
    d = _tconv_desc(gy, Cb, KT, dils, stride, ws, T_in, T_out, dh, dcoff)
    mc = mask.c()
    d.mask, d.center = C.pointer(mc), _ptr(center)
    lib = _lib_()
    nparts = lib.tamgcn_tconv_nparts(C.byref(d), 1)
    if nparts <= 0:
        raise RuntimeError('tamgcn_tconv_bwd: no tiling for this shape')
    part = empty(2, dh.shape[1], nparts, like=dh)
    d.stats_part, d.stats_ctot = _ptr(part), dh.shape[1]
    _lib.check(lib.tamgcn_tconv_bwd(C.byref(d), _stream()), 'tamgcn_tconv_bwd')
    return part


def tconv_wgrad_pays(V, Cb, stride):
    """Where the one-launch weight gradient of the temporal branches beats the per-branch launches (tools/tconv_bench.py,
    profiles/r04_tconv_bench.txt, 256 clips): 64-channel branches (129 vs 152 us, strided 176 vs 211), V = 25 except the
    16-channel stride-1 branches (0.9-1.35 ms vs 1.6-2.6 ms), V = 64 (1.30 vs 1.83 ms); at 16 / 32 channels and V = 20 the two
    are level (69 vs 74, 104 vs 104, 132 vs 126 us) and the per-branch launches stay."""
    if V > 32:
        return True
    if V % 4:
        return Cb >= 32 or stride > 1
    return Cb >= 64


def tconv_wgrad(gy, src, Cb, KT, dils, stride):
    """Weight gradients of every temporal branch in one launch: [dW_b (Cb, Cb, KT, 1) for b in branches].  gy: S over the
    gradient w.r.t. the concatenated output (two-source prologue), src: S over the forward's source (BatchNorm + ReLU)."""
    N, _, T_in, V = src.x1.shape
    T_out = gy.x1.shape[2]
    if V % 4:
        gy = S(with_slack(gy.x1), with_slack(gy.x2), gy.coef, gy.coff, gy.act)
        src = S(with_slack(src.x1), None, src.coef, src.coff, src.act)
    nb = len(dils)
    d = TconvDesc()
    d.src = gy.c()
    d.N, d.T_in, d.T_out, d.V, d.Cb, d.nb, d.KT, d.stride = N, T_in, T_out, V, Cb, nb, KT, stride
    for b, dl in enumerate(dils):
        d.dil[b] = int(dl)
    mc = src.c()
    d.mask = C.pointer(mc)
    lib = _lib_()
    mx = lib.tamgcn_tconv_wgrad_max_split(C.byref(d))
    if mx <= 0:
        raise RuntimeError('tamgcn_tconv_wgrad: no tiling for this shape')
    blocks = nb * (1 if Cb == 16 else (Cb // 32) ** 2)
    nsplit = max(1, min(mx, WGRAD_BLOCKS * 2 // blocks))
    part = empty(nsplit, nb, Cb, Cb, KT, like=gy.x1)
    d.y, d.yctot, d.ycoff = _ptr(part), nsplit, 0
    _lib.check(lib.tamgcn_tconv_wgrad(C.byref(d), _stream()), 'tamgcn_tconv_wgrad')
    return reduce_sum(part, nsplit, chunks=[(Cb, Cb, KT, 1)] * nb)


WGRAD_BLOCKS = int(os.environ.get('TAMGCN_WGRAD_BLOCKS', '512'))    # workgroups a weight gradient aims at (tiles x splits)


def wgrad(gy, src, M, K, KT=1, dil=1, stride=1, pad=0, rows=None):
    """Returns dW (M, K, KT, 1); with rows = [m0, m1, ...] (summing to M) a list of separate (m_i, K, KT, 1) tensors."""
    N, _, T_out, V = gy.x1.shape
    T_in = src.x1.shape[2]
    per_clip = max(gy.x1.shape[1] * T_out * V, src.x1.shape[1] * T_in * V)
    chunks = n_chunks(N, per_clip) if N * per_clip > CHUNK_ELEMS else [(0, N)]
    # same tile rule as wgrad_tile() in csrc/conv.hip: aim at ~4 resident workgroups per CU
    taps = KT > 1 and stride == 1 and T_in == T_out and (M > 32 or K > 32 or (V % 4 != 0 and (M > 16 or K > 16)))      # one window per tap on the LDS-DMA kernel
    if KT == 1 or taps:
        bm, bk = (64 if M <= 64 else 128), (64 if K <= 64 else 128)
    elif KT == 9:
        bm = bk = 32
    else:
        bm = bk = 32 if (M <= 32 and K <= 32) else 64
    lib = _lib_()
    tiles = ((M + bm - 1) // bm) * ((K + bk - 1) // bk) * (KT if taps else 1)
    descs = []
    for n0, n1 in chunks:
        d = WgradDesc()
        g_, s_ = (gy, src) if len(chunks) == 1 else (_slice_src(gy, n0, n1), _slice_src(src, n0, n1))
        d.gy, d.src = g_.c(), s_.c()
        d.N, d.M, d.K, d.T_in, d.T_out, d.V = n1 - n0, M, K, T_in, T_out, V
        d.KT, d.dil, d.stride, d.pad = KT, dil, stride, pad
        descs.append((d, g_, s_))
    per = max(1, (WGRAD_BLOCKS + tiles - 1) // tiles // len(chunks))
    splits = [max(1, min(lib.tamgcn_wgrad_max_split(C.byref(d)), per)) for d, _, _ in descs]
    nsplit = sum(splits)                                   # every chunk's partial slabs sit in ONE array: one reduction
    part = empty(nsplit, M, K, KT, like=gy.x1)
    off = 0
    for (d, _, _), ns in zip(descs, splits):
        d.part, d.nsplit = part.data_ptr() + 4 * off * M * K * KT, ns
        _lib.check(lib.tamgcn_wgrad(C.byref(d), _stream()), 'tamgcn_wgrad')
        off += ns
    if rows is not None:
        return reduce_sum(part, nsplit, chunks=[(m, K, KT, 1) for m in rows])
    if nsplit == 1:
        return part.view(M, K, KT, 1)
    return reduce_sum(part, nsplit).view(M, K, KT, 1)


class ReduceBatch:
    """with ReduceBatch(): every reduce_sum() inside only registers its slabs and returns the (not yet filled) output;
    ONE tamgcn_reduce_multi launch on the stream current at exit fills them all.  The caller guarantees that nothing
    inside reads a reduced value and that every producer has been joined into that stream before the exit.

    The active batch is per THREAD: nn.DataParallel (the reference's multi-GPU mode, processor/io.py:86-87) runs one
    autograd thread per device and ctypes / torch calls release the GIL, so a process-global would let one device's
    reductions land in another device's batch."""
    _tls = threading.local()

    @staticmethod
    def active():
        return getattr(ReduceBatch._tls, 'active', None)

    def __enter__(self):
        self.items, self.prev = [], ReduceBatch.active()
        ReduceBatch._tls.active = self
        return self

    def __exit__(self, *exc):
        ReduceBatch._tls.active = self.prev
        if self.items and exc[0] is None:
            arr = (ReduceDesc * len(self.items))()
            for i, (part, nsplit, stride, off, count, scale, acc, out) in enumerate(self.items):
                arr[i] = ReduceDesc(part.data_ptr() + 4 * off, _ptr(out), nsplit, int(acc), stride, count, scale)
            _lib.check(_lib_().tamgcn_reduce_multi(arr, len(self.items), _stream()), 'tamgcn_reduce_multi')
        self.items = []
        return False


def reduce_sum(part, nsplit, scale=1.0, out=None, accumulate=False, chunks=None, immediate=False):
    """part [nsplit][...] -> sum over the leading dim (deferred to the enclosing ReduceBatch, if any).

    chunks = list of shapes: the reduced vector is delivered as that many SEPARATE tensors (consecutive pieces), so that
    gradients of parameters a kernel treats as one packed operand leave as tensors autograd can adopt without a copy."""
    count = part.numel() // nsplit
    if chunks is not None:
        outs, off = [], 0
        flat = part.view(nsplit, count)
        for shp in chunks:
            n = 1
            for d_ in shp:
                n *= d_
            o = torch.empty(shp, device=part.device, dtype=torch.float32)
            _reduce_piece(flat, nsplit, count, off, n, scale, o)
            outs.append(o)
            off += n
        assert off == count, 'reduce_sum: chunk shapes do not cover the slab'
        return outs
    if out is None:
        out = torch.empty(part.shape[1:], device=part.device, dtype=torch.float32)
    _reduce_piece(part, nsplit, count, 0, count, scale, out, accumulate, immediate)   # immediate: the value is read inside the batch
    return out


def _reduce_piece(part, nsplit, stride, off, count, scale, out, accumulate=False, immediate=False):
    rb = None if immediate else ReduceBatch.active()
    if rb is not None:
        rb.items.append((part, nsplit, stride, off, count, scale, accumulate, out))   # keeps `part` alive until the launch
        return
    src = C.c_void_p(part.data_ptr() + 4 * off)
    if nsplit > 128:                                       # the two-stage kernel reduces in place: on a private copy of the piece
        tmp = part.view(nsplit, stride)[:, off:off + count].contiguous()
        _lib.check(_lib_().tamgcn_reduce_sum(_ptr(tmp), nsplit, count, count, scale, int(accumulate), _ptr(out), _stream()),
                   'tamgcn_reduce_sum')
        return
    _lib.check(_lib_().tamgcn_reduce_sum(src, nsplit, stride, count, scale, int(accumulate), _ptr(out), _stream()),
               'tamgcn_reduce_sum')


# ---------------------------------------------------------------------------
def coef_diff(coef_d, coef_y, mode):
    """[3][C] prologue coefficients of diff = down(x) - bn(y) in one launch (include/tamgcn.h: tamgcn_coef_diff)."""
    out = torch.empty_like(coef_y)
    _lib.check(_lib_().tamgcn_coef_diff(_ptr(coef_d), _ptr(coef_y), _ptr(out), coef_y.shape[1], mode, _stream()), 'tamgcn_coef_diff')
    return out


# ---------------------------------------------------------------------------
def bn_fwd_finalize(part, part_coff, count, gamma, beta, rmean, rvar, nbt, momentum, eps, training,
                    coef, save, coff, C_):
    lib = _lib_()
    if part is not None:
        pctot, nparts = part.shape[1], part.shape[2]
    else:
        pctot, nparts = 0, 0
    _lib.check(lib.tamgcn_bn_fwd_finalize(_ptr(part), pctot, part_coff, nparts, float(count),
                                          _ptr(gamma), _ptr(beta), _ptr(rmean), _ptr(rvar), _ptr(nbt),
                                          momentum, eps, int(training), _ptr(coef), _ptr(save),
                                          coef.shape[1], coff, C_, _stream()), 'tamgcn_bn_fwd_finalize')


class BNBatch:
    """Collects BatchNorm finalisations whose partial sums are ready together and issues them as ONE launch per
    direction on the stream current at flush() (tamgcn_bn_*_finalize_multi)."""

    def __init__(self):
        self.f, self.b, self.keep = [], [], []

    def fwd(self, part, part_coff, count, gamma, beta, rmean, rvar, nbt, momentum, eps, training, coef, save, coff, C_):
        d = _lib.BnFwdDesc()
        d.part = _ptr(part)
        d.part_ctot, d.nparts = (part.shape[1], part.shape[2]) if part is not None else (0, 0)
        d.part_coff, d.count = part_coff, float(count)
        d.gamma, d.beta, d.running_mean, d.running_var, d.num_batches_tracked = _ptr(gamma), _ptr(beta), _ptr(rmean), _ptr(rvar), _ptr(nbt)
        d.momentum, d.eps, d.training = momentum, eps, int(training)
        d.coef, d.save, d.coef_ctot, d.coef_coff, d.C = _ptr(coef), _ptr(save), coef.shape[1], coff, C_
        self.f.append(d)
        self.keep += [part, coef, save]

    def bwd(self, part, part_coff, count, gamma, save, save_coff, training, dgamma, dbeta, dbias, coef, coff, C_):
        d = _lib.BnBwdDesc()
        d.part, d.part_ctot, d.part_coff, d.nparts, d.count = _ptr(part), part.shape[1], part_coff, part.shape[2], float(count)
        d.gamma, d.save, d.save_ctot, d.save_coff, d.training = _ptr(gamma), _ptr(save), save.shape[1], save_coff, int(training)
        d.dgamma, d.dbeta, d.dbias_conv = _ptr(dgamma), _ptr(dbeta), _ptr(dbias)
        d.coef, d.coef_ctot, d.coef_coff, d.C = _ptr(coef), coef.shape[1], coff, C_
        self.b.append(d)
        self.keep += [part, save, coef, dgamma, dbeta, dbias]

    def flush(self):
        lib = _lib_()
        if self.f:
            arr = (_lib.BnFwdDesc * len(self.f))(*self.f)
            _lib.check(lib.tamgcn_bn_fwd_finalize_multi(arr, len(self.f), _stream()), 'tamgcn_bn_fwd_finalize_multi')
        if self.b:
            arr = (_lib.BnBwdDesc * len(self.b))(*self.b)
            _lib.check(lib.tamgcn_bn_bwd_finalize_multi(arr, len(self.b), _stream()), 'tamgcn_bn_bwd_finalize_multi')
        self.f, self.b, self.keep = [], [], []


def bn_bwd_finalize(part, part_coff, count, gamma, save, save_coff, training, dgamma, dbeta, dbias, coef, coff, C_):
    lib = _lib_()
    _lib.check(lib.tamgcn_bn_bwd_finalize(_ptr(part), part.shape[1], part_coff, part.shape[2], float(count),
                                          _ptr(gamma), _ptr(save), save.shape[1], save_coff, int(training),
                                          _ptr(dgamma), _ptr(dbeta), _ptr(dbias), _ptr(coef), coef.shape[1], coff, C_,
                                          _stream()), 'tamgcn_bn_bwd_finalize')


# ---------------------------------------------------------------------------
def tmean(src, C_):
    N, _, T, V = src.x1.shape
    xbar = empty(C_, N, V, like=src.x1)
    sc = src.c()
    _lib.check(_lib_().tamgcn_tmean(C.byref(sc), N, C_, T, V, _ptr(xbar), _stream()), 'tamgcn_tmean')
    return xbar


def _ctrgc_desc(x, pq, w3, b3, w4, b4, A, alpha, Cin, Cout, S, R, E=None):
    N, _, T, V = x.x1.shape
    d = CtrgcDesc()
    d.N, d.Cin, d.Cout, d.S, d.R, d.T, d.V = N, Cin, Cout, S, R, T, V
    d.x = x.c()
    d.pq, d.w3, d.b3, d.w4, d.b4, d.A, d.alpha = (_ptr(pq), _ptr(w3), _ptr(b3), _ptr(w4), _ptr(b4), _ptr(A), _ptr(alpha))
    d.E = _ptr(E)
    return d


def ctrgc_route(V):
    """'fused'  V = 20: LDS-resident E tiles, x3 GEMM + VALU aggregation in one kernel;
    'stream' V = 25: x3 = W3 x through the pointwise GEMM (kept in HBM), aggregation / dE accumulation on MFMA with the joints padded
             to 32 in LDS, E and the dE tail from the per-(n, subset) kernels of the fused family;
    'tiled'  V in {32, 64}: the same with tiled E / tail kernels (E of one channel is 48 KB at V = 64)."""
    k = _lib_().tamgcn_ctrgc_tiled_supported(int(V))
    if k == 1:
        return 'tiled'
    if k == 2 and V != 20:
        return 'stream'
    if V == 20:
        return 'fused'
    raise RuntimeError(f'tam_gcn_amd: CTRGC is built for V in {{20, 25, 32, 64}} joints, got V = {V}')


def ctrgc_tiled(V):
    """True when CTRGC runs as x3 GEMM + MFMA aggregation kernels (routes 'tiled' and 'stream')."""
    return ctrgc_route(V) != 'fused'


def ctrgc_build_E(x, pq, w3, b3, w4, b4, A, alpha, Cin, Cout, S, R):
    """E (N, S, Cout, V, V) for every channel, once per layer; hand it to ctrgc_fwd / ctrgc_bwd_dx3."""
    N, _, T, V = x.x1.shape
    if R > 32 or R % 4:
        raise RuntimeError(f'tam_gcn_amd: CTRGC with R = {R} rel-channels is not built (multiples of 4 up to 32: in_channels <= 256 + 7)')
    d = _ctrgc_desc(x, pq, w3, b3, w4, b4, A, alpha, Cin, Cout, S, R)
    E = empty(N, S, Cout, V, V, like=x.x1)
    if ctrgc_route(V) == 'tiled':
        _lib.check(_lib_().tamgcn_ctrgc_tiled_build_e(C.byref(d), _ptr(E), _stream()), 'tamgcn_ctrgc_tiled_build_e')
    else:
        _lib.check(_lib_().tamgcn_ctrgc_build_e(C.byref(d), _ptr(E), _stream()), 'tamgcn_ctrgc_build_e')
    return E


def _x3_gemm(x, w3, b3, Cin, Cout, S):
    """x3 = conv3(x) of every subset as ONE pointwise GEMM: (N, S*Cout, T, V)."""
    x3, _ = conv(x, K=Cin, w=w3, bias=b3.reshape(-1), M=S * Cout)
    return x3


def ctrgc_fwd(x, pq, w3, b3, w4, b4, A, alpha, Cin, Cout, S, R, stats, keep_x3=False, E=None):
    """Returns (y, stats_part, x3); x3 (N,S*Cout,T,V) only when keep_x3 (for the backward)."""
    N, _, T, V = x.x1.shape
    d = _ctrgc_desc(x, pq, w3, b3, w4, b4, A, alpha, Cin, Cout, S, R, E)
    y = empty(N, Cout, T, V, like=x.x1)
    part = empty(2, Cout, N, like=x.x1) if stats else None
    if ctrgc_tiled(V):
        if E is None:
            E = ctrgc_build_E(x, pq, w3, b3, w4, b4, A, alpha, Cin, Cout, S, R)
        x3 = _x3_gemm(x, w3, b3, Cin, Cout, S)
        chunks = n_chunks(N, S * Cout * T * V)
        parts = []
        for n0, n1 in chunks:                              # one launch unless x3 has >= 2^31 elements
            d.N = n1 - n0
            pc = None
            if stats:
                pc = part if len(chunks) == 1 else empty(2, Cout, n1 - n0, like=x.x1)
                parts.append(pc)
            _lib.check(_lib_().tamgcn_ctrgc_tiled_agg_fwd(C.byref(d), _ptr(x3[n0:n1]), _ptr(E[n0:n1]), _ptr(y[n0:n1]), _ptr(pc), _stream()),
                       'tamgcn_ctrgc_tiled_agg_fwd')
        if stats and len(chunks) > 1:
            part = torch.cat(parts, dim=2)
        return y, part, (x3 if keep_x3 else None)
    if E is None:
        E = ctrgc_build_E(x, pq, w3, b3, w4, b4, A, alpha, Cin, Cout, S, R)
        d.E = _ptr(E)
    x3 = empty(N, S * Cout, T, V, like=x.x1) if keep_x3 else None
    _lib.check(_lib_().tamgcn_ctrgc_fwd(C.byref(d), _ptr(y), _ptr(part), _ptr(x3), _stream()), 'tamgcn_ctrgc_fwd')
    return y, part, x3


def ctrgc_bwd_dx3(x, pq, w3, b3, w4, b4, A, alpha, Cin, Cout, S, R, dy, E=None, per_subset=False):
    """dx3 (N,S*Cout,T,V) = E^T . dy, and db3 [S*Cout]."""
    N, _, T, V = x.x1.shape
    d = _ctrgc_desc(x, pq, w3, b3, w4, b4, A, alpha, Cin, Cout, S, R, E)
    dyc = dy.c()
    dx3 = empty(N, S * Cout, T, V, like=x.x1)
    db3_part = empty(N, S * Cout, like=x.x1)
    if ctrgc_tiled(V):
        if E is None:
            E = ctrgc_build_E(x, pq, w3, b3, w4, b4, A, alpha, Cin, Cout, S, R)
        for n0, n1 in n_chunks(N, S * Cout * T * V):
            d.N = n1 - n0
            dyc = _slice_src(dy, n0, n1).c()
            _lib.check(_lib_().tamgcn_ctrgc_tiled_agg_bwd(C.byref(d), C.byref(dyc), _ptr(E[n0:n1]), _ptr(dx3[n0:n1]), _ptr(db3_part[n0:n1]),
                                                          _stream()), 'tamgcn_ctrgc_tiled_agg_bwd')
    else:
        if E is None:
            E = ctrgc_build_E(x, pq, w3, b3, w4, b4, A, alpha, Cin, Cout, S, R)
            d.E = _ptr(E)
        _lib.check(_lib_().tamgcn_ctrgc_bwd_dx3(C.byref(d), C.byref(dyc), _ptr(dx3), _ptr(db3_part), _stream()),
                   'tamgcn_ctrgc_bwd_dx3')
    return dx3, (reduce_sum(db3_part, N, chunks=[(Cout,)] * S) if per_subset else reduce_sum(db3_part, N))


def ctrgc_bwd_de(x, pq, w3, b3, w4, b4, A, alpha, Cin, Cout, S, R, dy, x3=None, per_subset=False):
    """The dE chain: dA [S,V,V], dw4 [S,Cout,R], db4 [S,Cout], dalpha [1], dpq [S*2*R,N,V]: a streaming accumulation of
    dE from dy and the x3 ctrgc_fwd kept (recomputed by one pointwise GEMM if it was not), then one per-(n, s) tail launch."""
    N, _, T, V = x.x1.shape
    d = _ctrgc_desc(x, pq, w3, b3, w4, b4, A, alpha, Cin, Cout, S, R)
    dyc = dy.c()
    like = x.x1
    route = ctrgc_route(V)
    dE = None
    if route != 'fused':
        if x3 is None:
            x3 = _x3_gemm(x, w3, b3, Cin, Cout, S)
        dE = empty(N, S, Cout, V, V, like=like)
        for n0, n1 in n_chunks(N, S * Cout * T * V):
            d.N = n1 - n0
            dyc = _slice_src(dy, n0, n1).c()
            _lib.check(_lib_().tamgcn_ctrgc_tiled_de_acc(C.byref(d), C.byref(dyc), _ptr(x3[n0:n1]), _ptr(dE[n0:n1]), _stream()),
                       'tamgcn_ctrgc_tiled_de_acc')
        d.N = N
    if route == 'tiled':
        NUC = _lib_().tamgcn_ctrgc_tiled_chunks(V)
        dA_part = empty(N, S, V, V, like=like)
        dw4_part = empty(N * NUC, S, Cout, R, like=like)
        db4_part = empty(N * NUC, S, Cout, like=like)
        dal_part = empty(N * S * NUC, 1, like=like)
        dpq = empty(NUC, S * 2 * R, N, V, like=like)
        _lib.check(_lib_().tamgcn_ctrgc_tiled_de_tail(C.byref(d), _ptr(dE), _ptr(dA_part), _ptr(dw4_part), _ptr(db4_part), _ptr(dal_part),
                                                      _ptr(dpq), _stream()), 'tamgcn_ctrgc_tiled_de_tail')
        ps = per_subset
        return (reduce_sum(dA_part, N), reduce_sum(dw4_part, N * NUC, chunks=[(Cout, R, 1, 1)] * S if ps else None),
                reduce_sum(db4_part, N * NUC, chunks=[(Cout,)] * S if ps else None), reduce_sum(dal_part, N * S * NUC),
                reduce_sum(dpq, NUC, immediate=True))
    if R > 32:
        raise RuntimeError('tam_gcn_amd: CTRGC with R > 32 rel-channels is not built')
    if dE is None:
        if x3 is None:                                      # the caller did not keep x3: one more pointwise GEMM
            x3 = _x3_gemm(x, w3, b3, Cin, Cout, S)
        dE = empty(N, S, Cout, V, V, like=like)
        _lib.check(_lib_().tamgcn_ctrgc_bwd_de_acc(C.byref(d), C.byref(dyc), _ptr(x3), _ptr(dE), _stream()),
                   'tamgcn_ctrgc_bwd_de_acc')
    G = 1                                               # channel groups per (n, subset); the per-workgroup fixed cost (D fill,
    #                                                     dp/dq sums) equals ~1.4 channel chunks, so splitting did not pay (measured)
    dA_part = empty(N * G, S, V, V, like=like)
    dw4_part = empty(N, S, Cout, R, like=like)
    db4_part = empty(N, S, Cout, like=like)
    dal_part = empty(N * S * G, 1, like=like)
    dpq = empty(G, S * 2 * R, N, V, like=like)
    _lib.check(_lib_().tamgcn_ctrgc_bwd_de_tail(C.byref(d), _ptr(dE), _ptr(dA_part), _ptr(dw4_part), _ptr(db4_part),
                                                _ptr(dal_part), _ptr(dpq), G, _stream()), 'tamgcn_ctrgc_bwd_de_tail')
    dpq = dpq[0]
    ps = per_subset
    return (reduce_sum(dA_part, N * G), reduce_sum(dw4_part, N, chunks=[(Cout, R, 1, 1)] * S if ps else None),
            reduce_sum(db4_part, N, chunks=[(Cout,)] * S if ps else None), reduce_sum(dal_part, N * S * G), dpq)


def ctrgc_bwd(x, pq, w3, b3, w4, b4, A, alpha, Cin, Cout, S, R, dy, x3=None, E=None):
    """Returns dx3 (N,S*Cout,T,V), db3 [S*Cout], dA [S,V,V], dw4 [S,Cout,R], db4 [S,Cout], dalpha [1], dpq [S*2*R,N,V]."""
    dx3, db3 = ctrgc_bwd_dx3(x, pq, w3, b3, w4, b4, A, alpha, Cin, Cout, S, R, dy, E=E)
    dA, dw4, db4, dal, dpq = ctrgc_bwd_de(x, pq, w3, b3, w4, b4, A, alpha, Cin, Cout, S, R, dy, x3)
    return dx3, db3, dA, dw4, db4, dal, dpq


# ---------------------------------------------------------------------------
def gcn_tail_fwd(y, o, res):
    N, _, T, V = y.x1.shape
    Cc = y.ctot
    g = empty(N, Cc, T, V, like=y.x1)
    yc, oc = y.c(), o.c()
    rc = res.c() if res is not None else None
    _lib.check(_lib_().tamgcn_gcn_tail_fwd(C.byref(yc), C.byref(oc), C.byref(rc) if rc is not None else None,
                                           N, Cc, T, V, _ptr(g), _stream()), 'tamgcn_gcn_tail_fwd')
    return g


def gcn_tail_bwd(dg, g, o, o_save):
    N, Cc, T, V = g.shape
    dsum, doz = empty_like(g), empty_like(g)
    part = empty(2, Cc, N, like=g)
    oc = o.c()
    _lib.check(_lib_().tamgcn_gcn_tail_bwd(_ptr(dg), _ptr(g), C.byref(oc), _ptr(o_save), N, Cc, T, V, _ptr(dsum), _ptr(doz),
                                           _ptr(part), _stream()), 'tamgcn_gcn_tail_bwd')
    return dsum, doz, part


def gcn_mid_bwd(dsum, ddiff, y_pre, y_save, r_pre, r_save, want_dres):
    N, Cc, T, V = dsum.shape
    dyb = empty_like(dsum)
    dres = empty_like(dsum) if want_dres else None
    part = empty(4 if r_pre is not None else 2, Cc, N, like=dsum)
    _lib.check(_lib_().tamgcn_gcn_mid_bwd(_ptr(dsum), _ptr(ddiff), _ptr(y_pre), _ptr(y_save), _ptr(r_pre), _ptr(r_save), N, Cc, T, V,
                                          _ptr(dyb), _ptr(dres), _ptr(part), _stream()), 'tamgcn_gcn_mid_bwd')
    return dyb, dres, part


def maxpool_fwd(src, C_, stride, y, ycoff, stats):
    N, _, T_in, V = src.x1.shape
    T_out = y.shape[2]
    part = empty(2, y.shape[1], N, like=y) if stats else None
    sc = src.c()
    _lib.check(_lib_().tamgcn_maxpool_fwd(C.byref(sc), N, C_, T_in, V, stride, _ptr(y), y.shape[1], ycoff, T_out,
                                          _ptr(part), _stream()), 'tamgcn_maxpool_fwd')
    return part


def maxpool_post_fwd(src, C_, stride, y, ycoff, coef, add, relu):
    """Eval-mode pooled branch finished in place: y[:, ycoff:ycoff+C_] = act(c1 * maxpool(src) + c0 + add)."""
    N, _, T_in, V = src.x1.shape
    sc = src.c()
    _lib.check(_lib_().tamgcn_maxpool_post_fwd(C.byref(sc), N, C_, T_in, V, stride, _ptr(y), y.shape[1], ycoff, y.shape[2],
                                               _ptr(coef), _ptr(add), int(relu), _stream()), 'tamgcn_maxpool_post_fwd')


def maxpool_bwd(gy, src, src_save, C_, stride, d, dcoff):
    N, _, T_in, V = src.x1.shape
    T_out = gy.x1.shape[2]
    part = empty(2, d.shape[1], N, like=d)
    gc, sc = gy.c(), src.c()
    _lib.check(_lib_().tamgcn_maxpool_bwd(C.byref(gc), C.byref(sc), _ptr(src_save), N, C_, T_in, T_out, V, stride, _ptr(d),
                                          d.shape[1], dcoff, _ptr(part), _stream()), 'tamgcn_maxpool_bwd')
    return part


XBAR_FOLD = os.environ.get('TAMGCN_XBAR_FOLD', '1') != '0'    # 0: every block computes its own frame means (tamgcn_tmean)


def add_act_fwd(a, res, relu, C_, rowmean=False, xbar=False):
    """rowmean: also return the (N, C) means over (t, v) of the output (the model head's pooling input).
    xbar: also return the (C, N, V) means over t of the output (the next block's pooled joint-embedding input) -- or None
    where the fused form does not apply (V % 4 != 0, V > 64)."""
    N, _, T, V = a.x1.shape
    out = empty(N, C_, T, V, like=a.x1)
    rm = torch.empty(N, C_, device=out.device, dtype=torch.float32) if rowmean else None
    xb = None
    if xbar and not rowmean and XBAR_FOLD and V % 4 == 0 and V <= 64:
        xb = empty(C_, N, V, like=a.x1)
    ac = a.c()
    rc = res.c() if res is not None else None
    _lib.check(_lib_().tamgcn_add_act_fwd(C.byref(ac), C.byref(rc) if rc is not None else None, int(relu),
                                          N, C_, T, V, _ptr(out), _ptr(rm), _ptr(xb), _stream()), 'tamgcn_add_act_fwd')
    if xbar:
        return out, xb
    return (out, rm) if rowmean else out


def add_act_bwd(dout, out, relu, a_pre, a_save, r_pre, r_save, want_dz):
    N, Cc, T, V = dout.shape
    dz = empty_like(dout) if want_dz else None
    part = empty(4 if r_pre is not None else 2, Cc, N, like=dout)
    _lib.check(_lib_().tamgcn_add_act_bwd(_ptr(dout), _ptr(out), int(relu), _ptr(a_pre), _ptr(a_save), _ptr(r_pre), _ptr(r_save), N, Cc, T, V,
                                          _ptr(dz), _ptr(part), _stream()), 'tamgcn_add_act_bwd')
    return dz, part


def apply(src, C_, y=None, ycoff=0):
    N, _, T, V = src.x1.shape
    if y is None:
        y = empty(N, C_, T, V, like=src.x1)
    sc = src.c()
    _lib.check(_lib_().tamgcn_apply(C.byref(sc), N, C_, T, V, _ptr(y), y.shape[1], ycoff, _stream()), 'tamgcn_apply')
    return y


# ---------------------------------------------------------------------------
# stem / head of Model
def stem_stats(x5, dout=None, center=None):
    """x5 (N, C, T, V, M) -> partial sums [2][C*V*M][N] for the data_bn finalize (forward: moments; backward with dout)."""
    N, C_, T, V, M = x5.shape
    part = empty(2, C_ * V * M, N, like=x5)
    _lib.check(_lib_().tamgcn_stem_stats(_ptr(x5), _ptr(dout), _ptr(center), N, C_, T, V, M, _ptr(part), _stream()), 'tamgcn_stem_stats')
    return part


def stem_apply(x5, coef, dout=None):
    """forward: (N*M, C, T, V) = c1*x + c0 (permuted);  with dout: dx (N, C, T, V, M) = c1*dout + c2*x + c0."""
    N, C_, T, V, M = x5.shape
    out = empty(N * M, C_, T, V, like=x5) if dout is None else torch.empty_like(x5)
    _lib.check(_lib_().tamgcn_stem_apply(_ptr(x5), _ptr(dout), _ptr(coef), N, C_, T, V, M, _ptr(out), _stream()), 'tamgcn_stem_apply')
    return out


def head_pool_fwd(x, M):
    NM, C_, T, V = x.shape
    pooled = empty(NM // M, C_, like=x)
    _lib.check(_lib_().tamgcn_head_pool_fwd(_ptr(x), NM // M, C_, T, V, M, _ptr(pooled), _stream()), 'tamgcn_head_pool_fwd')
    return pooled


def head_pool_bwd(dpooled, M, T, V):
    N, C_ = dpooled.shape
    dx = empty(N * M, C_, T, V, like=dpooled)
    _lib.check(_lib_().tamgcn_head_pool_bwd(_ptr(dpooled), N, C_, T, V, M, _ptr(dx), _stream()), 'tamgcn_head_pool_bwd')
    return dx


def head_fc_fwd(pooled, W, b):
    N, C_ = pooled.shape
    K = W.shape[0]
    logits = empty(N, K, like=pooled)
    _lib.check(_lib_().tamgcn_head_fc_fwd(_ptr(pooled), _ptr(W), _ptr(b), N, C_, K, _ptr(logits), _stream()), 'tamgcn_head_fc_fwd')
    return logits


def head_fc_bwd(dlogits, pooled, W):
    N, C_ = pooled.shape
    K = W.shape[0]
    dW, db, dpooled = torch.empty_like(W), empty(K, like=W), torch.empty_like(pooled)
    _lib.check(_lib_().tamgcn_head_fc_bwd(_ptr(dlogits), _ptr(pooled), _ptr(W), N, C_, K, _ptr(dW), _ptr(db), _ptr(dpooled), _stream()),
               'tamgcn_head_fc_bwd')
    return dW, db, dpooled


# ---------------------------------------------------------------------------
# input side: skeleton streams, the feeder's per-sample transform
STREAM_MODES = {'joint': 0, 'bone': 1, 'motion': 2, 'joint_motion': 2, 'bone_motion': 3}


def stream_derive(x5, parent, mode):
    """x5 (N, C, T, V, M) joint clips -> the 'bone' / 'motion' / 'bone_motion' stream (same shape); parent int32 [V]."""
    N, C_, T, V, M = x5.shape
    m = STREAM_MODES[mode] if isinstance(mode, str) else int(mode)
    if m == 0:
        return x5
    out = torch.empty_like(x5)
    _lib.check(_lib_().tamgcn_stream_derive(_ptr(x5), N, C_, T, V, M, _ptr(parent), m, _ptr(out), _stream()), 'tamgcn_stream_derive')
    return out


def feeder_transform(raw, offsets, rot, idx, parent, V, time_steps, center_joint, mode):
    """raw (sum L, V, 3) fp64, offsets int64 [N+1], rot fp64 (N, 3, 3), idx int32 (N, time_steps) -> (N, 3, time_steps, V, 1) fp32."""
    N = offsets.numel() - 1
    m = STREAM_MODES[mode] if isinstance(mode, str) else int(mode)
    out = torch.empty(N, 3, time_steps, V, 1, device=raw.device, dtype=torch.float32)
    _lib.check(_lib_().tamgcn_feeder_transform(_ptr(raw), _ptr(offsets), _ptr(rot), _ptr(idx), _ptr(parent), N, V, time_steps,
                                               center_joint, m, _ptr(out), _stream()), 'tamgcn_feeder_transform')
    return out


# ---------------------------------------------------------------------------
# loss of the harness step
def ce_fwd(logits, labels):
    N, K = logits.shape
    loss = torch.empty((), device=logits.device, dtype=torch.float32)
    g = torch.empty_like(logits)
    _lib.check(_lib_().tamgcn_ce_fwd(_ptr(logits), _ptr(labels), N, K, _ptr(loss), _ptr(g), _stream()), 'tamgcn_ce_fwd')
    return loss, g


def score_fuse(scores, weights, softmax, labels=None):
    """scores (S, N, K) f32, weights (S,) f32 -> fused (N, K), pred (N,) int64, class_stats (K, 2) int32 | None."""
    S_, N, K = scores.shape
    fused = torch.empty(N, K, device=scores.device, dtype=torch.float32)
    pred = torch.empty(N, device=scores.device, dtype=torch.int64)
    stats = torch.empty(K, 2, device=scores.device, dtype=torch.int32) if labels is not None else None
    _lib.check(_lib_().tamgcn_score_fuse(_ptr(scores), _ptr(weights), S_, N, K, int(bool(softmax)), _ptr(labels), _ptr(fused), _ptr(pred),
                                         _ptr(stats), _stream()), 'tamgcn_score_fuse')
    return fused, pred, stats


def ce_bwd(g, dloss):
    N, K = g.shape
    dl = torch.empty_like(g)
    _lib.check(_lib_().tamgcn_ce_bwd(_ptr(g), _ptr(dloss), N, K, _ptr(dl), _stream()), 'tamgcn_ce_bwd')
    return dl
