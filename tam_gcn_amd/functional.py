"""Forward/backward orchestration of the HIP ops for the CTR-GCN block, and the
autograd.Function wrappers the nn.Module mirror (models/ctrgcn.py) calls.

Everything numeric happens in libtamgcn.so; this file only sequences kernels,
allocates outputs and routes gradients to parameters.  Train-mode BatchNorm is
split as  producer(+moment partials) -> bn_*_finalize -> consumer(prologue).

Reference arithmetic being reproduced: models/ctrgcn.py:52-69 (TemporalConv),
:72-147 (MultiScale_TemporalConv), :150-177 (CTRGC), :179-193 (unit_tcn),
:196-263 (unit_gcn), :266-284 (TCN_GCN_unit).
"""
import contextlib
import os
import threading

import torch

from . import ops
from .ops import S, RELU


# ---------------------------------------------------------------------------
# Side streams.  Weight-gradient kernels are off the backward critical path and the TCN
# branches are mutually independent; most of these kernels are latency-bound with 1-2
# workgroups per CU, so letting them overlap raises utilisation.  fork()/join() use
# wait_stream (event record + wait), which HIP graph capture turns into parallel branches.
# ---------------------------------------------------------------------------
_SIDE = {}
_SIDE_LOCK = threading.Lock()
USE_SIDE_STREAMS = os.environ.get('TAMGCN_SIDE_STREAMS', '1') != '0'

def _side_streams(device, main, k):
    """The k side streams that belong to MAIN stream `main` of `device`.

    The pool is keyed by the main stream, not only by the device: a tensor allocated while side stream S is current
    belongs to S's block pool in torch's caching allocator and is handed out again to the next allocation on S as soon
    as Python drops it.  That is safe only if S's next work is ordered after every consumer of the old tensor, which
    fork() guarantees for consumers on the main stream S was forked from -- and for no other stream.  Two callers on
    two main streams (four models on four streams, GraphedForward beside a training step, two DataParallel threads on
    one device) therefore never share a side stream.  torch hands out streams from a pool of 32 per device that wraps
    around; if a fresh stream aliases one already in use here, this main stream gets no side streams (its branches run
    in order on the main stream): slower, never a race."""
    key = (device.index, main.cuda_stream, k)
    got = _SIDE.get(key)
    if got is None:
        with _SIDE_LOCK:
            got = _SIDE.get(key)
            if got is None:
                taken = {main.cuda_stream}
                for (d, m, _), ss in _SIDE.items():
                    if d == device.index:
                        taken.add(m)
                        taken.update(s.cuda_stream for s in ss)
                got = []
                for _ in range(k):
                    s = torch.cuda.Stream(device)
                    if s.cuda_stream in taken:
                        got = []
                        break
                    taken.add(s.cuda_stream)
                    got.append(s)
                _SIDE[key] = got
    return got


# Streams a caller runs whole models on, beside other models on other streams (model_stream()).  Their blocks' branches run
# in order on that stream: the concurrency comes from the models, and HIP (ROCm 7.2) faults inside hipStreamEndCapture --
# a segmentation fault, not an error code -- when a stream that JOINED a capture forks further streams (a two-level
# fork: capture stream -> model stream -> side streams), while one-level forks capture fine.  Found by bisection
# (tools/stream_capture_check.py, profiles/r03_stream_capture_bisect.txt): plain torch modules on two forked streams
# capture; this model on a forked stream captures with the side streams off and faults with them on, forward alone included.
_MODEL_STREAMS = set()
_CAPTURE_ORIGINS = set()


@contextlib.contextmanager
def model_stream(stream):
    """``with model_stream(s): loss = criterion(model(x), y)`` -- run a model (and, through autograd, its backward) on
    stream ``s`` beside other models on other streams.  Like ``torch.cuda.stream(s)``, and registers ``s`` so that the
    blocks under it do not fork side streams of their own (see _MODEL_STREAMS)."""
    _MODEL_STREAMS.add((stream.device.index, stream.cuda_stream))
    with torch.cuda.stream(stream):
        yield stream


def allow_side_streams_in_capture(stream):
    """Declare ``stream`` the ORIGIN of a HIP-graph capture (``torch.cuda.graph(g, stream=stream)``): blocks then fork their
    side streams from it during the capture.  torch's default capture stream is recognised without this."""
    _CAPTURE_ORIGINS.add((stream.device.index, stream.cuda_stream))


def _side_ok(device, main):
    if not USE_SIDE_STREAMS:
        return False
    key = (device.index, main.cuda_stream)
    if key in _MODEL_STREAMS:
        return False
    if torch.cuda.is_current_stream_capturing():
        # Only the stream the capture began on may fork (one level).  That stream cannot be asked from HIP; torch's
        # own capture stream and the declared origins are known, anything else runs its branches in order.
        dcs = getattr(torch.cuda.graph, 'default_capture_stream', None)
        if not ((dcs is not None and dcs.cuda_stream == main.cuda_stream) or key in _CAPTURE_ORIGINS):
            return False
    return True


class Fork:
    """with Fork(device, k) as f:  f.on(i) -> context running on side stream i; joins on exit."""

    def __init__(self, device, k=2):
        self.main = torch.cuda.current_stream(device)
        self.side = _side_streams(device, self.main, k) if _side_ok(device, self.main) else []
        self.used = set()

    def on(self, i):
        if not self.side:
            return contextlib.nullcontext()
        st = self.side[i % len(self.side)]
        if i % len(self.side) not in self.used:
            st.wait_stream(self.main)                  # fork: everything enqueued so far on main is visible
            self.used.add(i % len(self.side))
        return torch.cuda.stream(st)

    def refork(self):
        """Call after enqueuing more work on main that later side work depends on."""
        for i in self.used:
            self.side[i].wait_stream(self.main)

    def mark(self, i):
        """Event after what side stream i has been given so far; main.wait_event(it) later joins THAT point only."""
        if not self.side or (i % len(self.side)) not in self.used:
            return None
        ev = torch.cuda.Event()
        ev.record(self.side[i % len(self.side)])
        return ev

    def wait(self, ev):
        if ev is not None:
            self.main.wait_event(ev)

    def join(self, i):
        """main waits for side stream i now (its results are needed on main)."""
        if self.side and (i % len(self.side)) in self.used:
            self.main.wait_stream(self.side[i % len(self.side)])

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        for i in self.used:
            self.main.wait_stream(self.side[i])        # join
        self.used.clear()
        return False


class BN:
    """Borrowed view of one nn.BatchNorm{1,2}d's tensors."""
    __slots__ = ('w', 'b', 'rm', 'rv', 'nbt', 'mom', 'eps', 'C', 'm')

    def __init__(self, m):
        if m.momentum is None or not m.track_running_stats or not m.affine:
            raise RuntimeError('tam_gcn_amd: BatchNorm must be affine with running stats and a fixed momentum')
        self.w, self.b, self.rm, self.rv, self.nbt = m.weight, m.bias, m.running_mean, m.running_var, m.num_batches_tracked
        self.mom, self.eps, self.C, self.m = float(m.momentum), float(m.eps), m.num_features, m

    def tensors(self):
        return (self.w, self.b, self.rm, self.rv)

    def fwd(self, part, part_coff, count, training, coef, save, coff, batch=None):
        """batch (ops.BNBatch): only registered; batch.flush() launches it with the others."""
        if training:                                       # the kernel rewrites the running statistics through raw pointers:
            _bn_epoch(self.m)[0] += 1                      # Tensor._version does not see it, the eval-coefficient cache must
        if batch is not None:
            batch.fwd(part, part_coff, count, self.w, self.b, self.rm, self.rv, self.nbt, self.mom, self.eps, training, coef, save, coff, self.C)
            return
        ops.bn_fwd_finalize(part, part_coff, count, self.w, self.b, self.rm, self.rv, self.nbt, self.mom, self.eps,
                            training, coef, save, coff, self.C)

    def bwd(self, part, part_coff, count, save, save_coff, training, coef, coff, want_dbias=False, batch=None):
        dev = self.w.device
        dg = torch.empty(self.C, device=dev)
        db = torch.empty(self.C, device=dev)
        dbias = torch.empty(self.C, device=dev) if want_dbias else None
        if batch is not None:
            batch.bwd(part, part_coff, count, self.w, save, save_coff, training, dg, db, dbias, coef, coff, self.C)
        else:
            ops.bn_bwd_finalize(part, part_coff, count, self.w, save, save_coff, training, dg, db, dbias, coef, coff, self.C)
        return dg, db, dbias


def _bn_epoch(m):
    """One-element list counting the train-mode statistic updates of BatchNorm module `m`.  A mutable object, created
    when the owning module is built (tag_batchnorms_), so that nn.DataParallel's replicas -- whose __dict__ is a shallow
    copy of the original's -- share it: a replica's training step invalidates the original's eval-coefficient cache."""
    ep = m.__dict__.get('_tamgcn_epoch')
    if ep is None:
        ep = m.__dict__['_tamgcn_epoch'] = [0]
    return ep


def tag_batchnorms_(module):
    """Give every BatchNorm under `module` its update counter now (called at the end of the mirror modules' __init__)."""
    for m in module.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            _bn_epoch(m)
    return module


def _coef(C_, like):
    return torch.empty(3, C_, device=like.device), torch.empty(2, C_, device=like.device)


# ---------------------------------------------------------------------------
# Eval mode without autograd (the inference callers -- cross-modal attention, ensemble eval, visual.py).
# BatchNorm is a per-channel affine of the running statistics there: its coefficients depend only on four tensors per
# BatchNorm, so they are computed once and cached on the module until one of those tensors changes (in-place updates bump
# Tensor._version, .to() changes data_ptr); the 96 finalize launches of a forward disappear.  The convolution epilogues
# then finish their consumers' work (tamgcn_conv post_coef / post_act): every MS-TCN branch writes relu(bn(branch) + residual)
# straight into its channel slice of the block output -- no cat_pre tensor, no add_act pass.
# ---------------------------------------------------------------------------
EVAL_FUSED = os.environ.get('TAMGCN_EVAL_FUSED', '1') != '0'


def _eval_cached(owner, tag, bns, build):
    """build() -> value, cached on nn.Module `owner` under `tag` while the tensors of the BN views `bns` are unchanged."""
    key = tuple((t.data_ptr(), t._version) for bn in bns for t in bn.tensors()) + tuple(_bn_epoch(bn.m)[0] for bn in bns)
    arena = getattr(bns[0].w, '_tamgcn_arena', None) if bns else None      # weight / bias may be views of a ParamArena's flat
    if arena is not None:                                                  # buffer: its updates do not bump their _version
        key += arena.state_version()
    cache = owner.__dict__.setdefault('_tamgcn_eval_cache', {})
    hit = cache.get(tag)
    if hit is not None and hit[0] == key:
        return hit[1]
    val = build()
    cache[tag] = (key, val)
    return val


def _eval_coefs(placed, Ctot, like):
    """coef [3][Ctot] / save [2][Ctot] of eval-mode BatchNorms placed at channel offsets: [(BN, coff), ...]."""
    coef, save = _coef(Ctot, like)
    for bn, coff in placed:
        bn.fwd(None, 0, 1, False, coef, save, coff)
    return coef, save


# ===========================================================================
# unit_gcn
# ===========================================================================
class GcnParams:
    """Tensors of one unit_gcn, with the per-subset conv weights stacked."""
    __slots__ = ('S', 'R', 'Cin', 'Cout', 'PA', 'alpha', 'W12', 'B12', 'W3', 'B3', 'W4', 'B4',
                 'bn', 'mode', 'Wd', 'bd', 'bnd', 'Wo', 'bo', 'bno')


def gcn_forward(x, P, training, save):
    N, Cin, T, V = x.shape
    S_, R, Cout = P.S, P.R, P.Cout
    count = N * T * V
    xs = S(x)
    fk = Fork(x.device, 2)
    fk.__enter__()
    d_pre = coef_d = save_d = None
    ev = None                                          # eval without autograd: cached BatchNorm coefficients, no finalize launches
    if EVAL_FUSED and not training and not save:
        bns = [P.bn, P.bno] + ([P.bnd] if P.mode == 'conv' else [])

        def build():
            cy, sy = _eval_coefs([(P.bn, 0)], Cout, x)
            co, so = _eval_coefs([(P.bno, 0)], Cout, x)
            cd, sd_ = _eval_coefs([(P.bnd, 0)], Cout, x) if P.mode == 'conv' else (None, None)
            cdiff = ops.coef_diff(cd, cy, {'conv': 0, 'identity': 1}.get(P.mode, 2))
            return dict(y=(cy, sy), o=(co, so), d=(cd, sd_), diff=cdiff)
        ev = _eval_cached(P.bn.m, 'unit_gcn', bns, build)
    if P.mode == 'conv':                               # independent of the CTRGC chain: side stream
        with fk.on(0):
            d_pre, dpart = ops.conv(xs, K=Cin, w=P.Wd, bias=P.bd, M=Cout, stats=training)
            if ev is not None:
                coef_d, save_d = ev['d']
    # pooled joint embeddings (conv1/conv2 commute with the mean over T): the previous block's last pass left
    # them beside its output (ops.add_act_fwd(xbar=True) -> attach_xbar) unless something touched the tensor since
    xbar = taken_xbar(x, Cin)
    if xbar is None:
        xbar = ops.tmean(xs, Cin)                                         # (Cin, N, V)
    pq, _ = ops.conv(S(xbar.view(1, Cin, N, V)), K=Cin, w=P.W12, bias=P.B12, M=S_ * 2 * R)
    pq = pq.view(S_ * 2 * R, N, V)
    # E = alpha*(W4 tanh(p_u - q_v) + b4) + A for every channel, once per layer (N*S*Cout*V*V floats: 1.7 GB over the model at
    # 256 clips); the forward and the dx3 backward load their tiles of it.  x3 = conv3(x) leaves the forward kernel too
    # (its tile is in LDS anyway; 3 activations per block) for the backward's dE accumulation and conv3 weight gradient.
    E = ops.ctrgc_build_E(xs, pq, P.W3, P.B3, P.W4, P.B4, P.PA, P.alpha, Cin, Cout, S_, R)
    y_pre, ypart, x3 = ops.ctrgc_fwd(xs, pq, P.W3, P.B3, P.W4, P.B4, P.PA, P.alpha, Cin, Cout, S_, R, stats=training,
                                     keep_x3=save, E=E)
    fk.__exit__()                                      # join: d_pre is needed now
    if ev is not None:
        coef_y, save_y = ev['y']
    else:                                              # bn and down's BatchNorm in one launch
        bb = ops.BNBatch()
        coef_y, save_y = _coef(Cout, x)
        P.bn.fwd(ypart, 0, count, training, coef_y, save_y, 0, batch=bb)
        if P.mode == 'conv':
            coef_d, save_d = _coef(Cout, x)
            P.bnd.fwd(dpart, 0, count, training, coef_d, save_d, 0, batch=bb)
        bb.flush()
    if P.mode == 'conv':
        res = S(d_pre, coef=coef_d)
        coef_diff = ev['diff'] if ev is not None else ops.coef_diff(coef_d, coef_y, 0)
        diff = S(d_pre, y_pre, coef_diff)
    elif P.mode == 'identity':
        res = xs
        coef_diff = ev['diff'] if ev is not None else ops.coef_diff(None, coef_y, 1)
        diff = S(x, y_pre, coef_diff)
    else:                                                                  # residual=False: down(x) = 0
        res = None
        coef_diff = ev['diff'] if ev is not None else ops.coef_diff(None, coef_y, 2)
        diff = S(y_pre, None, coef_diff)
    o_pre, opart = ops.conv(diff, K=Cout, w=P.Wo, bias=P.bo, M=Cout, stats=training)
    if ev is not None:
        coef_o, save_o = ev['o']
    else:
        coef_o, save_o = _coef(Cout, x)
        P.bno.fwd(opart, 0, count, training, coef_o, save_o, 0)
    g = ops.gcn_tail_fwd(S(y_pre, coef=coef_y), S(o_pre, coef=coef_o), res)
    sv = None
    if save:
        sv = dict(x=x, xbar=xbar, pq=pq, x3=x3, E=E, y_pre=y_pre, d_pre=d_pre, o_pre=o_pre, g=g, coef_y=coef_y, save_y=save_y,
                  coef_d=coef_d, save_d=save_d, coef_o=coef_o, save_o=save_o, coef_diff=coef_diff,
                  training=training)
    return g, sv


def gcn_backward(P, sv, dg, need_dx=True, extra_dx=None):
    """Returns (dx, grads) with grads keyed like GcnParams' tensors.  All partial-slab reductions of the weight and
    parameter gradients are deferred into ONE launch after the side streams have joined (ops.ReduceBatch)."""
    with ops.ReduceBatch():
        return _gcn_backward(P, sv, dg, need_dx, extra_dx)


def _gcn_backward(P, sv, dg, need_dx=True, extra_dx=None):
    x, xbar, pq, y_pre, d_pre, o_pre, g = sv['x'], sv['xbar'], sv['pq'], sv['y_pre'], sv['d_pre'], sv['o_pre'], sv['g']
    training = sv['training']
    N, Cin, T, V = x.shape
    S_, R, Cout = P.S, P.R, P.Cout
    count = N * T * V
    G = {}
    fk = Fork(x.device, 2)
    fk.__enter__()
    # tail: relu, tanh(BN(offset conv))
    dsum, doz, part_o = ops.gcn_tail_bwd(dg, g, S(o_pre, coef=sv['coef_o']), sv['save_o'])
    coefb_o = torch.empty(3, Cout, device=x.device)
    G['bno.w'], G['bno.b'], G['bo'] = P.bno.bwd(part_o, 0, count, sv['save_o'], 0, training, coefb_o, 0, want_dbias=True)
    gyo = S(doz, o_pre, coefb_o)
    ddiff, _ = ops.conv(gyo, K=Cout, w=P.Wo, bias=None, M=Cout, wmode=1)
    if P.mode == 'conv':
        diff = S(d_pre, y_pre, sv['coef_diff'])
    elif P.mode == 'identity':
        diff = S(x, y_pre, sv['coef_diff'])
    else:
        diff = S(y_pre, None, sv['coef_diff'])
    with fk.on(0):
        G['Wo'] = ops.wgrad(gyo, diff, M=Cout, K=Cout)
    dyb, dres, part2 = ops.gcn_mid_bwd(dsum, ddiff, y_pre, sv['save_y'], d_pre if P.mode == 'conv' else None,
                                       sv['save_d'], want_dres=P.mode != 'zero')
    coefb_y = torch.empty(3, Cout, device=x.device)
    bb = ops.BNBatch()                                 # bn and down's BatchNorm: both from gcn_mid_bwd's moments, one launch
    G['bn.w'], G['bn.b'], _ = P.bn.bwd(part2, 0, count, sv['save_y'], 0, training, coefb_y, 0, batch=bb)
    coefb_d = None
    if P.mode == 'conv':
        coefb_d = torch.empty(3, Cout, device=x.device)
        G['bnd.w'], G['bnd.b'], G['bd'] = P.bnd.bwd(part2[2:4], 0, count, sv['save_d'], 0, training, coefb_d, 0,
                                                     want_dbias=True, batch=bb)
    bb.flush()
    dy = S(dyb, y_pre, coefb_y)
    # CTRGC
    xs = S(x)
    cargs = (xs, pq, P.W3, P.B3, P.W4, P.B4, P.PA, P.alpha, Cin, Cout, S_, R, dy)
    fk.refork()                                        # dyb / coefb_y are ready on main
    with fk.on(1):                                     # the dE chain runs beside the dx3 -> dx chain
        G['PA'], G['W4'], G['B4'], G['alpha'], dpq = ops.ctrgc_bwd_de(*cargs, x3=sv['x3'], per_subset=True)
        dpq4 = S(dpq.view(1, S_ * 2 * R, N, V))
        G['W12'] = ops.wgrad(dpq4, S(xbar.view(1, Cin, N, V)), M=S_ * 2 * R, K=Cin, rows=[R] * (2 * S_))   # one tensor per parameter
        G['B12'] = dpq.sum((1, 2))
        dxbar = None
        if need_dx:
            dxbar, _ = ops.conv(dpq4, K=S_ * 2 * R, w=P.W12, bias=None, M=Cin, wmode=1)    # (1, Cin, N, V)
    dx3, G['B3'] = ops.ctrgc_bwd_dx3(*cargs, E=sv['E'], per_subset=True)
    fk.refork()
    with fk.on(0):
        G['W3'] = ops.wgrad(S(dx3), xs, M=S_ * Cout, K=Cin, rows=[Cout] * S_)
    fk.join(1)                                         # dxbar feeds the dx conv
    dx = None
    if need_dx:
        dx, _ = ops.conv(S(dx3), K=S_ * Cout, w=P.W3, bias=None, M=Cin, wmode=1,
                         bcast=dxbar.view(Cin, N, V), bcast_scale=1.0 / T,
                         add1=dres if P.mode == 'identity' else None, add2=extra_dx)
    if P.mode == 'conv':
        gyd = S(dres, d_pre, coefb_d)
        fk.refork()
        with fk.on(0):
            G['Wd'] = ops.wgrad(gyd, xs, M=Cout, K=Cin)
        if need_dx:
            ops.conv(gyd, K=Cout, w=P.Wd, bias=None, M=Cin, wmode=1, add1=dx, y=dx)
    fk.__exit__()
    return dx, G


# ===========================================================================
# MultiScale_TemporalConv (+ optional residual and final ReLU of TCN_GCN_unit)
# ===========================================================================
class TcnParams:
    """nb dilated branches + max-pool branch + plain 1x1 branch.
    Win/bin: entry 1x1 convs of branches 0..nb stacked [(nb+1)*Cb, Cin];
    bn_in[b], Wt[b]/bt[b]/bn_t[b] (b < nb), bn_pool, Wl/bl/bn_l (last branch);
    residual: mode 'zero' | 'identity' | 'conv' with Wr/br/bnr, rk (kernel size)."""
    __slots__ = ('Cin', 'Cout', 'Cb', 'nb', 'ks', 'dils', 'stride', 'Win', 'bin', 'bn_in', 'Wt', 'bt', 'bn_t',
                 'bn_pool', 'Wl', 'bl', 'bn_l', 'rmode', 'Wr', 'br', 'bnr', 'rk', 'relu')


def attach_xbar(out, xb):
    """Remember the frame means (C, N, V) of `out` ON the tensor object, with the version they belong to."""
    if xb is not None:
        out._tamgcn_xbar = (xb, out._version)
    return out


def taken_xbar(x, Cin):
    """The frame means a producer attached to exactly this tensor object, if it has not been written since."""
    tag = getattr(x, '_tamgcn_xbar', None)
    if tag is None:
        return None
    xb, ver = tag
    N, C_, _, V = x.shape
    if ver != x._version or C_ != Cin or tuple(xb.shape) != (Cin, N, V) or xb.device != x.device:
        return None
    return xb


def _tpad(k, d):
    return (k + (k - 1) * (d - 1) - 1) // 2


def tcn_forward(g, P, training, save, xres=None, pool=None):
    """g (N,Cin,T,V) -> (N,Cout,T2,V).  xres: tensor the residual branch reads
    (defaults to g, as in MultiScale_TemporalConv; TCN_GCN_unit passes the block input).
    pool: None | list -- the final pass appends the (N, Cout) means over (t, v) of its output (the model head's
    pooling input, models/ctrgcn.py:343-345); left empty by the launch-fused eval path, whose output has no single
    final pass."""
    N, Cin, T, V = g.shape
    Cb, nb, s = P.Cb, P.nb, P.stride
    Cout = P.Cout
    T2 = (T - 1) // s + 1
    cnt1, cnt2 = N * T * V, N * T2 * V
    gs = S(g)
    Ch = (nb + 1) * Cb
    if EVAL_FUSED and not training and not save:
        return _tcn_forward_eval(g, P, xres), None
    fk = Fork(g.device, 2)
    fk.__enter__()
    cat_pre = ops.empty(N, Cout, T2, V, like=g)
    coef_c, save_c = _coef(Cout, g)
    bo = ops.BNBatch()                                 # the branches' output norms: one launch after the join
    with fk.on(1):                                     # the plain 1x1 branch only needs g
        _, lpart = ops.conv(gs, K=Cin, w=P.Wl, bias=P.bl, M=Cb, stride=s, y=cat_pre, ycoff=(nb + 1) * Cb, T_out=T2,
                            stats=training)
        P.bn_l.fwd(lpart, (nb + 1) * Cb, cnt2, training, coef_c, save_c, (nb + 1) * Cb, batch=bo)
    h_pre, hpart = ops.conv(gs, K=Cin, w=P.Win, bias=P.bin, M=Ch, stats=training)
    coef_h, save_h = _coef(Ch, g)
    bi = ops.BNBatch()                                 # the three entry norms: one launch
    for b in range(nb + 1):
        P.bn_in[b].fwd(hpart, b * Cb, cnt1, training, coef_h, save_h, b * Cb, batch=bi)
    bi.flush()
    fk.refork()                                        # h_pre and coef_h are ready on main
    if ops.tconv_supported(V, Cb, P.ks, P.dils, s, T):
        # every temporal branch and the pooled one in ONE launch (csrc/tconv.hip): h_pre is read once per branch slice
        part = ops.tconv_fwd(S(h_pre, coef=coef_h, act=RELU), Cb, P.ks[0], P.dils, s, P.Wt, P.bt, True, cat_pre, 0, stats=training)
        for b in range(nb):
            P.bn_t[b].fwd(part, b * Cb, cnt2, training, coef_c, save_c, b * Cb, batch=bo)
        P.bn_pool.fwd(part, nb * Cb, cnt2, training, coef_c, save_c, nb * Cb, batch=bo)
    else:
        for b in range(nb):
            k, d = P.ks[b], P.dils[b]
            with (fk.on(b) if b > 0 else contextlib.nullcontext()):        # branch 0 on main, the others beside it
                _, part = ops.conv(S(h_pre, coef=coef_h, coff=b * Cb, act=RELU), K=Cb, w=P.Wt[b], bias=P.bt[b], M=Cb,
                                   KT=k, dil=d, stride=s, pad=_tpad(k, d), y=cat_pre, ycoff=b * Cb, T_out=T2,
                                   stats=training)
                P.bn_t[b].fwd(part, b * Cb, cnt2, training, coef_c, save_c, b * Cb, batch=bo)
        with fk.on(0):
            part = ops.maxpool_fwd(S(h_pre, coef=coef_h, coff=nb * Cb, act=RELU), Cb, s, cat_pre, nb * Cb, stats=training)
            P.bn_pool.fwd(part, nb * Cb, cnt2, training, coef_c, save_c, nb * Cb, batch=bo)
    r_pre = coef_r = save_r = None
    if xres is None:
        xres = g
    if P.rmode == 'identity':
        res = S(xres)
    elif P.rmode == 'conv':
        rk = P.rk
        r_pre, rpart = ops.conv(S(xres), K=xres.shape[1], w=P.Wr, bias=P.br, M=Cout, KT=rk, stride=s,
                                pad=(rk - 1) // 2, T_out=T2, stats=training)
        coef_r, save_r = _coef(Cout, g)
        P.bnr.fwd(rpart, 0, cnt2, training, coef_r, save_r, 0, batch=bo)
        res = S(r_pre, coef=coef_r)
    else:
        res = None
    fk.__exit__()                                      # join all branches
    bo.flush()
    if pool is not None:
        out, rm = ops.add_act_fwd(S(cat_pre, coef=coef_c), res, P.relu, Cout, rowmean=True)
        pool.append(rm)
    else:
        # the next block's frame means ride along in train mode / under autograd (the step this path is built for); a plain
        # eval forward keeps tamgcn_tmean, so that it stays bit-compatible with the launch-fused eval path (EVAL_FUSED)
        if training or save:
            out, xb = ops.add_act_fwd(S(cat_pre, coef=coef_c), res, P.relu, Cout, xbar=True)
            attach_xbar(out, xb)
        else:
            out = ops.add_act_fwd(S(cat_pre, coef=coef_c), res, P.relu, Cout)
    sv = None
    if save:
        sv = dict(g=g, xres=xres, h_pre=h_pre, cat_pre=cat_pre, r_pre=r_pre, out=out, coef_h=coef_h, save_h=save_h,
                  coef_c=coef_c, save_c=save_c, coef_r=coef_r, save_r=save_r, training=training)
    return out, sv


def _tcn_forward_eval(g, P, xres):
    """Eval mode without autograd: five launches (entry convs, two temporal convs, pooled branch, plain branch) + one
    for a convolutional residual; every branch finishes relu(bn(branch) + residual) in its own epilogue."""
    N, Cin, T, V = g.shape
    Cb, nb, s, Cout = P.Cb, P.nb, P.stride, P.Cout
    T2 = (T - 1) // s + 1
    Ch = (nb + 1) * Cb
    gs = S(g)
    bns = list(P.bn_in) + list(P.bn_t) + [P.bn_pool, P.bn_l] + ([P.bnr] if P.rmode == 'conv' else [])

    def build():
        ch, _ = _eval_coefs([(P.bn_in[b], b * Cb) for b in range(nb + 1)], Ch, g)
        cc, _ = _eval_coefs([(P.bn_t[b], b * Cb) for b in range(nb)] + [(P.bn_pool, nb * Cb), (P.bn_l, (nb + 1) * Cb)], Cout, g)
        cr = _eval_coefs([(P.bnr, 0)], Cout, g)[0] if P.rmode == 'conv' else None
        return ch, cc, cr
    coef_h, coef_c, coef_r = _eval_cached(P.bn_l.m, 'ms_tcn', bns, build)
    if xres is None:
        xres = g
    res = None
    if P.rmode == 'identity':
        res = xres
    elif P.rmode == 'conv':
        rk = P.rk
        res, _ = ops.conv(S(xres), K=xres.shape[1], w=P.Wr, bias=P.br, M=Cout, KT=rk, stride=s, pad=(rk - 1) // 2, T_out=T2,
                          post_coef=coef_r)
    h_pre, _ = ops.conv(gs, K=Cin, w=P.Win, bias=P.bin, M=Ch)
    out = ops.empty(N, Cout, T2, V, like=g)
    fk = Fork(g.device, 2)
    fk.__enter__()
    with fk.on(1):
        ops.conv(gs, K=Cin, w=P.Wl, bias=P.bl, M=Cb, stride=s, y=out, ycoff=(nb + 1) * Cb, T_out=T2, post_coef=coef_c,
                 post_act=P.relu, add1=res)
    for b in range(nb):
        k, d = P.ks[b], P.dils[b]
        with (fk.on(b) if b > 0 else contextlib.nullcontext()):
            ops.conv(S(h_pre, coef=coef_h, coff=b * Cb, act=RELU), K=Cb, w=P.Wt[b], bias=P.bt[b], M=Cb, KT=k, dil=d, stride=s,
                     pad=_tpad(k, d), y=out, ycoff=b * Cb, T_out=T2, post_coef=coef_c, post_act=P.relu, add1=res)
    with fk.on(0):
        ops.maxpool_post_fwd(S(h_pre, coef=coef_h, coff=nb * Cb, act=RELU), Cb, s, out, nb * Cb, coef_c, res, P.relu)
    fk.__exit__()
    return out


def tcn_backward(P, sv, dout, need_dg=True, need_dxres=True):
    with ops.ReduceBatch():                            # one launch for every slab reduction of this backward
        return _tcn_backward(P, sv, dout, need_dg, need_dxres)


def _tcn_backward(P, sv, dout, need_dg=True, need_dxres=True):
    """Returns (dg, dxres, grads).  dxres is None for rmode 'zero'; for 'identity' it is the
    masked upstream gradient itself (caller adds it)."""
    g, xres, h_pre, cat_pre, r_pre, out = sv['g'], sv['xres'], sv['h_pre'], sv['cat_pre'], sv['r_pre'], sv['out']
    training = sv['training']
    N, Cin, T, V = g.shape
    Cb, nb, s, Cout = P.Cb, P.nb, P.stride, P.Cout
    T2 = cat_pre.shape[2]
    cnt1, cnt2 = N * T * V, N * T2 * V
    Ch = (nb + 1) * Cb
    G = {}
    fk = Fork(g.device, 2)
    fk.__enter__()
    dz, part = ops.add_act_bwd(dout, out, P.relu, cat_pre, sv['save_c'], r_pre, sv['save_r'], want_dz=bool(P.relu))
    if dz is None:
        dz = dout
    coefb_c = torch.empty(3, Cout, device=g.device)
    bo = ops.BNBatch()                                 # every output norm of the block (and the residual's) in one launch
    G['bn_t'] = []
    G['bt'] = []
    for b in range(nb):
        dgam, dbet, dbias = P.bn_t[b].bwd(part, b * Cb, cnt2, sv['save_c'], b * Cb, training, coefb_c, b * Cb, True, batch=bo)
        G['bn_t'].append((dgam, dbet))
        G['bt'].append(dbias)
    dgam, dbet, _ = P.bn_pool.bwd(part, nb * Cb, cnt2, sv['save_c'], nb * Cb, training, coefb_c, nb * Cb, batch=bo)
    G['bn_pool'] = (dgam, dbet)
    dgam, dbet, G['bl'] = P.bn_l.bwd(part, (nb + 1) * Cb, cnt2, sv['save_c'], (nb + 1) * Cb, training, coefb_c,
                                     (nb + 1) * Cb, True, batch=bo)
    G['bn_l'] = (dgam, dbet)
    coefb_r = None
    if P.rmode == 'conv':
        coefb_r = torch.empty(3, Cout, device=g.device)
        dgam, dbet, G['br'] = P.bnr.bwd(part[2:4], 0, cnt2, sv['save_r'], 0, training, coefb_r, 0, True, batch=bo)
        G['bnr'] = (dgam, dbet)
    bo.flush()

    def gcat(coff):
        return S(dz, cat_pre, coefb_c, coff=coff)

    dh = ops.empty(N, Ch, T, V, like=g)
    coefb_h = torch.empty(3, Ch, device=g.device)
    G['Wt'] = []
    G['bn_in'] = []
    dbin = []
    # temporal branches: branch 0's data gradient on main, the others' beside it (small, latency-bound launches);
    # every branch's weight gradient follows on a side stream
    marks = []
    bi = ops.BNBatch()                                 # the entry norms' backward: one launch once every dh slice is written
    fused = ops.tconv_supported(V, Cb, P.ks, P.dils, s, T)
    fused_w = fused and ops.tconv_wgrad_pays(V, Cb, s)
    if fused:                                          # every temporal branch's data gradient in ONE launch (csrc/tconv.hip)
        hp_all = ops.tconv_bwd(gcat(0), Cb, P.ks[0], P.dils, s, P.Wt, S(h_pre, coef=sv['coef_h']), sv['save_h'], dh, 0)
    for b in range(nb):
        k, d = P.ks[b], P.dils[b]
        pad = _tpad(k, d)
        if fused:
            hp = hp_all
        else:
            with (fk.on(b) if b > 0 else contextlib.nullcontext()):
                _, hp = ops.conv(gcat(b * Cb), K=Cb, w=P.Wt[b], bias=None, M=Cb, KT=k, dil=d, stride=1,
                                 pad=(k - 1) * d - pad, wmode=1, up=s, y=dh, ycoff=b * Cb, T_out=T,
                                 mask=S(h_pre, coef=sv['coef_h'], coff=b * Cb), aux=h_pre, aux_center=sv['save_h'], auxcoff=b * Cb,
                                 stats=True)
        dgam, dbet, dbias = P.bn_in[b].bwd(hp, b * Cb, cnt1, sv['save_h'], b * Cb, training, coefb_h, b * Cb, True, batch=bi)
        if b > 0 and not fused:
            marks.append(fk.mark(b))                   # dh slice and moments of branch b done (its wgrad is not awaited)
        G['bn_in'].append((dgam, dbet))
        dbin.append(dbias)
        if not fused_w:
            with fk.on(b):
                G['Wt'].append(ops.wgrad(gcat(b * Cb), S(h_pre, coef=sv['coef_h'], coff=b * Cb, act=RELU), M=Cb, K=Cb,
                                         KT=k, dil=d, stride=s, pad=pad))
    if fused_w:                                        # every branch's weight gradient in ONE launch, beside the main chain
        with fk.on(0):
            G['Wt'] = ops.tconv_wgrad(gcat(0), S(h_pre, coef=sv['coef_h'], act=RELU), Cb, P.ks[0], P.dils, s)
    hp = ops.maxpool_bwd(gcat(nb * Cb), S(h_pre, coef=sv['coef_h'], coff=nb * Cb, act=RELU), sv['save_h'], Cb, s, dh,
                         nb * Cb)
    dgam, dbet, dbias = P.bn_in[nb].bwd(hp, nb * Cb, cnt1, sv['save_h'], nb * Cb, training, coefb_h, nb * Cb, True, batch=bi)
    G['bn_in'].append((dgam, dbet))
    dbin.append(dbias)
    for ev in marks:
        fk.wait(ev)                                    # the other branches' slices of dh and their moments
    bi.flush()
    G['bin'] = dbin                                    # per branch (separate tensors: autograd adopts them without a copy)
    gs = S(g)
    gyh = S(dh, h_pre, coefb_h)
    fk.refork()                                        # dh and its BN-backward coefficients are ready
    with fk.on(0):
        G['Win'] = ops.wgrad(gyh, gs, M=Ch, K=Cin, rows=[Cb] * (nb + 1))
    with fk.on(1):
        G['Wl'] = ops.wgrad(gcat((nb + 1) * Cb), gs, M=Cb, K=Cin, stride=s)
    dg = None
    if need_dg:
        dg, _ = ops.conv(gyh, K=Ch, w=P.Win, bias=None, M=Cin, wmode=1)
        ops.conv(gcat((nb + 1) * Cb), K=Cb, w=P.Wl, bias=None, M=Cin, wmode=1, y=dg, T_out=T2, ostride=s, add1=dg)
    dxres = None
    if P.rmode == 'identity':
        dxres = dz
    elif P.rmode == 'conv':
        gyr = S(dz, r_pre, coefb_r)
        rk = P.rk
        rpad = (rk - 1) // 2
        fk.refork()
        with fk.on(1):
            G['Wr'] = ops.wgrad(gyr, S(xres), M=Cout, K=xres.shape[1], KT=rk, stride=s, pad=rpad)
        if need_dxres:
            Tx = xres.shape[2]
            if rk == 1:
                dxres = ops.zeros_like(xres) if s > 1 else ops.empty_like(xres)
                ops.conv(gyr, K=Cout, w=P.Wr, bias=None, M=xres.shape[1], wmode=1, y=dxres, T_out=T2, ostride=s)
            else:
                dxres, _ = ops.conv(gyr, K=Cout, w=P.Wr, bias=None, M=xres.shape[1], KT=rk, stride=1,
                                    pad=(rk - 1) - rpad, wmode=1, up=s, T_out=Tx)
    fk.__exit__()
    return dg, dxres, G


# ===========================================================================
# autograd wrappers.  Parameter order is fixed by the *_tensors() helpers of
# the modules (models/ctrgcn.py); cfg objects carry everything non-differentiable.
# ===========================================================================
# Autograd runs Function.forward with grad mode off, and ctx.needs_input_grad reports the inputs' requires_grad flags even
# when the CALLER is under torch.no_grad(): the caller's grad mode is recorded by run() (what the modules call instead of
# apply()), so that inference saves nothing for a backward that cannot happen and takes the fused eval path (row f2).
_CALL = threading.local()


class _Fn(torch.autograd.Function):
    @classmethod
    def run(cls, *args):
        prev = getattr(_CALL, 'grad', True)
        _CALL.grad = torch.is_grad_enabled()
        try:
            return cls.apply(*args)
        finally:
            _CALL.grad = prev


def _needs(ctx):
    return getattr(_CALL, 'grad', True) and any(ctx.needs_input_grad)


class UnitGCNFn(_Fn):
    @staticmethod
    def forward(ctx, mod, x, *params):
        P = mod._pack(params)
        g, sv = gcn_forward(x, P, mod.training, save=_needs(ctx))
        ctx.mod, ctx.P, ctx.sv = mod, P, sv
        return g

    @staticmethod
    def backward(ctx, dg):
        dx, G = gcn_backward(ctx.P, ctx.sv, dg.contiguous(), need_dx=ctx.needs_input_grad[1])
        ctx.sv = None
        return (None, dx) + tuple(ctx.mod._route(G))


class MSTCNFn(_Fn):
    @staticmethod
    def forward(ctx, mod, x, *params):
        P = mod._pack(params)
        out, sv = tcn_forward(x, P, mod.training, save=_needs(ctx))
        ctx.mod, ctx.P, ctx.sv = mod, P, sv
        return out

    @staticmethod
    def backward(ctx, dout):
        need = ctx.needs_input_grad[1]
        dg, dxres, G = tcn_backward(ctx.P, ctx.sv, dout.contiguous(), need_dg=need, need_dxres=need)
        ctx.sv = None
        if need and dxres is not None:
            dg = dg + dxres            # stand-alone module with its own residual (rare path)
        return (None, dg) + tuple(ctx.mod._route(G))


class TCNGCNUnitFn(_Fn):
    """relu(tcn1(gcn1(x)) + residual(x)) as one autograd node (models/ctrgcn.py:282-284)."""

    @staticmethod
    def forward(ctx, mod, x, ngcn, emit, *params):
        """emit: also return the (N, Cout) row means of the output (non-differentiable side output: the model head's
        pooling input; its gradient reaches the block through `out`).  Empty (0, Cout) when the pass that would
        produce it did not run (launch-fused eval path)."""
        Pg = mod.gcn1._pack(params[:ngcn])
        Pt = mod._pack_tcn(params[ngcn:])
        save = _needs(ctx)
        training = mod.training
        g, svg = gcn_forward(x, Pg, training, save)
        pool = [] if emit else None
        out, svt = tcn_forward(g, Pt, training, save, xres=x, pool=pool)
        ctx.mod, ctx.Pg, ctx.Pt, ctx.svg, ctx.svt, ctx.ngcn = mod, Pg, Pt, svg, svt, ngcn
        if not emit:
            return out
        rm = pool[0] if pool else out.new_empty(0, out.shape[1])
        ctx.mark_non_differentiable(rm)
        return out, rm

    @staticmethod
    def backward(ctx, dout, _drm=None):
        need_dx = ctx.needs_input_grad[1]
        dg, dxres, Gt = tcn_backward(ctx.Pt, ctx.svt, dout.contiguous(), need_dg=True, need_dxres=need_dx)
        ctx.svt = None
        dx, Gg = gcn_backward(ctx.Pg, ctx.svg, dg, need_dx=need_dx, extra_dx=dxres)
        ctx.svg = None
        return (None, dx, None, None) + tuple(ctx.mod.gcn1._route(Gg)) + tuple(ctx.mod._route_tcn(Gt))


# ---------------------------------------------------------------------------
# stand-alone CTRGC (single subset; A and alpha are forward() arguments)
# ---------------------------------------------------------------------------
class CTRGCFn(_Fn):
    @staticmethod
    def forward(ctx, x, A, alpha, w1, b1, w2, b2, w3, b3, w4, b4):
        N, Cin, T, V = x.shape
        Cout, R = w3.shape[0], w1.shape[0]
        W12 = torch.cat((w1.reshape(R, Cin), w2.reshape(R, Cin)))
        B12 = torch.cat((b1, b2))
        W3, W4 = w3.reshape(Cout, Cin), w4.reshape(1, Cout, R)
        A3 = A.reshape(1, V, V).contiguous()
        al = alpha.reshape(1).to(torch.float32).contiguous()
        xs = S(x)
        xbar = ops.tmean(xs, Cin)
        pq, _ = ops.conv(S(xbar.view(1, Cin, N, V)), K=Cin, w=W12, bias=B12, M=2 * R)
        pq = pq.view(2 * R, N, V)
        save = _needs(ctx)
        E = ops.ctrgc_build_E(xs, pq, W3, b3, W4, b4.reshape(1, Cout), A3, al, Cin, Cout, 1, R)
        y, _, x3 = ops.ctrgc_fwd(xs, pq, W3, b3, W4, b4.reshape(1, Cout), A3, al, Cin, Cout, 1, R, stats=False, keep_x3=save, E=E)
        ctx.sv = (x, xbar, pq, W12, W3, b3, W4, b4.reshape(1, Cout), A3, al, E, x3) if save else None
        ctx.shapes = (A.shape, alpha.shape, w1.shape, w3.shape, w4.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, xbar, pq, W12, W3, b3, W4, b4, A3, al, E, x3 = ctx.sv
        ctx.sv = None
        N, Cin, T, V = x.shape
        Cout, R = W3.shape[0], W4.shape[2]
        xs = S(x)
        dx3, db3, dA, dW4, db4, dal, dpq = ops.ctrgc_bwd(xs, pq, W3, b3, W4, b4, A3, al, Cin, Cout, 1, R,
                                                         S(dy.contiguous()), x3=x3, E=E)
        dpq4 = S(dpq.view(1, 2 * R, N, V))
        dW12 = ops.wgrad(dpq4, S(xbar.view(1, Cin, N, V)), M=2 * R, K=Cin)
        dB12 = dpq.sum((1, 2))
        dW3 = ops.wgrad(S(dx3), xs, M=Cout, K=Cin)
        dx = None
        if ctx.needs_input_grad[0]:
            dxbar, _ = ops.conv(dpq4, K=2 * R, w=W12, bias=None, M=Cin, wmode=1)
            dx, _ = ops.conv(S(dx3), K=Cout, w=W3, bias=None, M=Cin, wmode=1, bcast=dxbar.view(Cin, N, V),
                             bcast_scale=1.0 / T)
        As, als, w1s, w3s, w4s = ctx.shapes
        return (dx, dA.reshape(As), dal.reshape(als), dW12[:R].reshape(w1s), dB12[:R], dW12[R:].reshape(w1s),
                dB12[R:], dW3.reshape(w3s), db3, dW4.reshape(w4s), db4.reshape(Cout))


# ---------------------------------------------------------------------------
# conv k x 1 + BatchNorm (TemporalConv, unit_tcn)
# ---------------------------------------------------------------------------
class ConvBNFn(_Fn):
    @staticmethod
    def forward(ctx, cfg, x, w, b, gamma, beta):
        k, s, d, pad, bnmod = cfg
        bn = BN(bnmod)
        training = bnmod.training
        N, Cin, T, V = x.shape
        M = w.shape[0]
        y_pre, part = ops.conv(S(x), K=Cin, w=w, bias=b, M=M, KT=k, dil=d, stride=s, pad=pad, stats=training)
        T2 = y_pre.shape[2]
        coef, save = _coef(M, x)
        bn.fwd(part, 0, N * T2 * V, training, coef, save, 0)
        out = ops.apply(S(y_pre, coef=coef), M)
        ctx.sv = (x, w, y_pre, save, bn, training, cfg)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w, y_pre, save, bn, training, cfg = ctx.sv
        k, s, d, pad, _ = cfg
        dout = dout.contiguous()
        N, Cin, T, V = x.shape
        M, T2 = y_pre.shape[1], y_pre.shape[2]
        _, part = ops.add_act_bwd(dout, None, 0, y_pre, save, None, None, want_dz=False)
        coefb = torch.empty(3, M, device=x.device)
        dgam, dbet, dbias = bn.bwd(part, 0, N * T2 * V, save, 0, training, coefb, 0, True)
        gy = S(dout, y_pre, coefb)
        dw = ops.wgrad(gy, S(x), M=M, K=Cin, KT=k, dil=d, stride=s, pad=pad)
        dx = None
        if ctx.needs_input_grad[1]:
            if k == 1:
                dx = ops.zeros_like(x) if s > 1 else ops.empty_like(x)
                ops.conv(gy, K=M, w=w, bias=None, M=Cin, wmode=1, y=dx, T_out=T2, ostride=s)
            else:
                dx, _ = ops.conv(gy, K=M, w=w, bias=None, M=Cin, KT=k, dil=d, stride=1, pad=(k - 1) * d - pad,
                                 wmode=1, up=s, T_out=T)
        return None, dx, dw, dbias, dgam, dbet


# ===========================================================================
# stem (data_bn + permutes) and head (global mean-pool + fc) of Model  -- SURVEY.md §8 row f1
# ===========================================================================
class StemFn(_Fn):
    """x (N, C, T, V, M) -> data_bn (channels (m, v, c), statistics over (n, t)) -> (N*M, C, T, V)
    (reference models/ctrgcn.py:328-332: two permute copies and a BatchNorm1d)."""

    @staticmethod
    def forward(ctx, bn_mod, x, w, b):
        x = x.contiguous()
        N, C_, T, V, M = x.shape
        bn = BN(bn_mod)
        J = C_ * V * M
        if bn.C != J:
            raise RuntimeError(f'tam_gcn_amd: data_bn has {bn.C} features, input gives {J}')
        training = bn_mod.training
        if EVAL_FUSED and not training and not _needs(ctx):
            coef, save = _eval_cached(bn_mod, 'stem', [bn], lambda: _eval_coefs([(bn, 0)], J, x))
        else:
            coef, save = torch.empty(3, J, device=x.device), torch.empty(2, J, device=x.device)
            part = ops.stem_stats(x) if training else None
            bn.fwd(part, 0, N * T, training, coef, save, 0)
        out = ops.stem_apply(x, coef)
        ctx.bn, ctx.training = bn, training
        ctx.save_for_backward(x, save)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, save = ctx.saved_tensors
        N, C_, T, V, M = x.shape
        dout = dout.contiguous()
        J = C_ * V * M
        part = ops.stem_stats(x, dout, save[0].contiguous())
        coefb = torch.empty(3, J, device=x.device)
        dg, db, _ = ctx.bn.bwd(part, 0, N * T, save, 0, ctx.training, coefb, 0)
        dx = ops.stem_apply(x, coefb, dout) if ctx.needs_input_grad[1] else None
        return None, dx, dg, db


class HeadFn(_Fn):
    """x10 (N*M, C, T, V) -> mean over (m, t, v) -> fc  (reference models/ctrgcn.py:343-348, drop_out = 0)."""

    @staticmethod
    def forward(ctx, x, W, b, M):
        x = x.contiguous()
        pooled = ops.head_pool_fwd(x, M)
        logits = ops.head_fc_fwd(pooled, W, b)
        ctx.M, ctx.TV = M, (x.shape[2], x.shape[3])
        ctx.save_for_backward(pooled, W)
        return logits

    @staticmethod
    def backward(ctx, dl):
        pooled, W = ctx.saved_tensors
        dW, db, dpooled = ops.head_fc_bwd(dl.contiguous(), pooled, W)
        dx = ops.head_pool_bwd(dpooled, ctx.M, *ctx.TV) if ctx.needs_input_grad[0] else None
        return dx, dW, db, None


# ===========================================================================
# loss of the harness step -- SURVEY.md §8 row f1 (reference processor/recognition_rgb.py:19, :62: nn.CrossEntropyLoss())
# ===========================================================================
class CrossEntropyFn(_Fn):
    @staticmethod
    def forward(ctx, logits, labels):
        logits = logits.contiguous()
        loss, g = ops.ce_fwd(logits, labels.contiguous())
        ctx.save_for_backward(g)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (g,) = ctx.saved_tensors
        return ops.ce_bwd(g, dloss.to(torch.float32).contiguous()), None


class CrossEntropyLoss(torch.nn.Module):
    """Drop-in for the ``nn.CrossEntropyLoss()`` the reference's processors build (default arguments: mean reduction, no
    class weights, no label smoothing, no ignore_index hits): two HIP launches instead of four aten kernels."""

    def forward(self, logits, labels):
        if logits.dim() != 2 or labels.dim() != 1 or labels.dtype != torch.int64 or logits.dtype != torch.float32:
            raise RuntimeError('tam_gcn_amd.CrossEntropyLoss: expected (N, K) fp32 logits and (N,) int64 class indices')
        if not logits.is_cuda:
            raise RuntimeError('tam_gcn_amd: the loss runs on MI355X only (got a CPU tensor); there is no CPU fallback')
        return torch.ops.tamgcn.cross_entropy(logits, labels)[0]        # tam_gcn_amd.torch_ops


# ===========================================================================
# ST-GCN block (reference models/stgcn.py:37-99) -- SURVEY.md §8 row f4.
# The spatial graph convolution  einsum('nkctv,kvw->nctw')  of conv(x) with the static A * edge_importance is CTRGC's
# aggregation with a topology that does not depend on the sample or the channel: E_k[n,c,u,v] = (A*imp)[k,v,u], i.e. the
# fused CTRGC kernels with alpha = 0 (E = alpha*(W4 tanh(.) + b4) + A = A).  conv3 plays the 1x1 conv (Cin -> K*Cout).
# ===========================================================================
def _stgcn_ctrgc_args(x, Ae, W3, B3, Cout):
    K, V = Ae.shape[0], Ae.shape[1]
    N = x.shape[0]
    dev = x.device
    R = 4                                               # smallest rel-channel count the kernels take; all-zero refinement
    z = torch.zeros
    return dict(pq=z(K * 2 * R, N, V, device=dev), w4=z(K, Cout, R, device=dev), b4=z(K, Cout, device=dev),
                A=Ae.transpose(1, 2).contiguous(), alpha=z(1, device=dev), R=R, K=K)


class StGcnFn(_Fn):
    """relu(tcn(gcn(x, A)) + residual(x)) of one st_gcn block as one autograd node."""

    @staticmethod
    def forward(ctx, mod, tail, x, Ae, w3, b3, g1, be1, wt, bt, g2, be2, *res_params):
        """tail False: stop at the second BatchNorm's output (no residual, no ReLU) -- st_gcn with an active Dropout."""
        N, Cin, T, V = x.shape
        Cout, s, kt = mod.out_channels, mod.stride, mod.t_kernel
        training = mod.training
        save = _needs(ctx)
        bn1, bn2 = BN(mod.tcn[0]), BN(mod.tcn[3])
        xs = S(x)
        W3, Ae = w3.reshape(w3.shape[0], Cin), Ae.contiguous()
        a = _stgcn_ctrgc_args(x, Ae, W3, b3, Cout)
        cargs = (xs, a['pq'], W3, b3, a['w4'], a['b4'], a['A'], a['alpha'], Cin, Cout, a['K'], a['R'])
        E = ops.ctrgc_build_E(*cargs)                    # alpha = 0: E[n, k, c] = (A * importance)[k]^T for every sample and channel
        y_pre, ypart, x3 = ops.ctrgc_fwd(*cargs, stats=training, keep_x3=save, E=E)
        cnt1 = N * T * V
        coef1, save1 = _coef(Cout, x)
        bn1.fwd(ypart, 0, cnt1, training, coef1, save1, 0)
        pad = (kt - 1) // 2
        z_pre, zpart = ops.conv(S(y_pre, coef=coef1, act=RELU), K=Cout, w=wt, bias=bt, M=Cout, KT=kt, stride=s, pad=pad, stats=training)
        T2 = z_pre.shape[2]
        cnt2 = N * T2 * V
        coef2, save2 = _coef(Cout, x)
        bn2.fwd(zpart, 0, cnt2, training, coef2, save2, 0)
        r_pre = coef_r = save_r = None
        rmode = mod._rmode if tail else 'zero'
        if rmode == 'identity':
            res = xs
        elif rmode == 'conv':
            wr, br = res_params[0], res_params[1]
            bnr = BN(mod.residual[1])
            r_pre, rpart = ops.conv(xs, K=Cin, w=wr, bias=br, M=Cout, KT=1, stride=s, pad=0, T_out=T2, stats=training)
            coef_r, save_r = _coef(Cout, x)
            bnr.fwd(rpart, 0, cnt2, training, coef_r, save_r, 0)
            res = S(r_pre, coef=coef_r)
        else:
            res = None
        out = ops.add_act_fwd(S(z_pre, coef=coef2), res, bool(tail), Cout)
        if save:
            ctx.sv = dict(x=x, Ae=Ae, W3=W3, b3=b3, a=a, x3=x3, E=E, y_pre=y_pre, z_pre=z_pre, r_pre=r_pre, out=out, coef1=coef1, save1=save1,
                          coef2=coef2, save2=save2, coef_r=coef_r, save_r=save_r, wt=wt, res_params=res_params, training=training,
                          rmode=rmode, tail=bool(tail))
        ctx.mod = mod
        return out

    @staticmethod
    def backward(ctx, dout):
        mod, sv = ctx.mod, ctx.sv
        ctx.sv = None
        x, W3, b3, a = sv['x'], sv['W3'], sv['b3'], sv['a']
        y_pre, z_pre, r_pre, out, wt = sv['y_pre'], sv['z_pre'], sv['r_pre'], sv['out'], sv['wt']
        training = sv['training']
        N, Cin, T, V = x.shape
        Cout, s, kt = mod.out_channels, mod.stride, mod.t_kernel
        T2 = z_pre.shape[2]
        cnt1, cnt2 = N * T * V, N * T2 * V
        pad = (kt - 1) // 2
        need_dx = ctx.needs_input_grad[2]
        rmode = sv['rmode']
        bn1, bn2 = BN(mod.tcn[0]), BN(mod.tcn[3])
        xs = S(x)
        with ops.ReduceBatch():
            dout = dout.contiguous()
            dz, part = ops.add_act_bwd(dout, out, int(sv['tail']), z_pre, sv['save2'], r_pre, sv['save_r'], want_dz=sv['tail'])
            if dz is None:
                dz = dout
            coefb2 = torch.empty(3, Cout, device=x.device)
            dg2, dbe2, dbt = bn2.bwd(part, 0, cnt2, sv['save2'], 0, training, coefb2, 0, want_dbias=True)
            gz = S(dz, z_pre, coefb2)
            h = S(y_pre, coef=sv['coef1'], act=RELU)
            dwt = ops.wgrad(gz, h, M=Cout, K=Cout, KT=kt, stride=s, pad=pad)
            dh, hp = ops.conv(gz, K=Cout, w=wt, bias=None, M=Cout, KT=kt, dil=1, stride=1, pad=(kt - 1) - pad, wmode=1, up=s, T_out=T,
                              mask=S(y_pre, coef=sv['coef1']), aux=y_pre, aux_center=sv['save1'], auxcoff=0, stats=True)
            coefb1 = torch.empty(3, Cout, device=x.device)
            dg1, dbe1, _ = bn1.bwd(hp, 0, cnt1, sv['save1'], 0, training, coefb1, 0)
            dy = S(dh, y_pre, coefb1)
            cargs = (xs, a['pq'], W3, b3, a['w4'], a['b4'], a['A'], a['alpha'], Cin, Cout, a['K'], a['R'], dy)
            dx3, db3 = ops.ctrgc_bwd_dx3(*cargs, E=sv['E'])
            dA, _, _, _, _ = ops.ctrgc_bwd_de(*cargs, x3=sv['x3'])
            dw3 = ops.wgrad(S(dx3), xs, M=a['K'] * Cout, K=Cin)
            gres = []
            add1 = None
            if rmode == 'identity':
                add1 = dz
            elif rmode == 'conv':
                wr = sv['res_params'][0]
                bnr = BN(mod.residual[1])
                coefb_r = torch.empty(3, Cout, device=x.device)
                dgr, dber, dbr = bnr.bwd(part[2:4], 0, cnt2, sv['save_r'], 0, training, coefb_r, 0, True)
                gyr = S(dz, r_pre, coefb_r)
                dwr = ops.wgrad(gyr, xs, M=Cout, K=Cin, KT=1, stride=s, pad=0)
                gres = [dwr, dbr, dgr, dber]
                if need_dx:
                    add1 = ops.zeros_like(x) if s > 1 else ops.empty_like(x)
                    ops.conv(gyr, K=Cout, w=wr, bias=None, M=Cin, wmode=1, y=add1, T_out=T2, ostride=s)
            dx = None
            if need_dx:
                dx, _ = ops.conv(S(dx3), K=a['K'] * Cout, w=W3, bias=None, M=Cin, wmode=1, add1=add1)
        dAe = dA.transpose(1, 2) if ctx.needs_input_grad[3] else None
        return (None, None, dx, dAe, dw3.reshape(a['K'] * Cout, Cin, 1, 1), db3, dg1, dbe1, dwt, dbt, dg2, dbe2, *gres)


class TemporalConvFn(_Fn):
    """Plain k x 1 convolution with bias over (N, C, T, V): stride, dilation and zero padding along T
    (nn.Conv2d(C, M, (k, 1), (s, 1), (p, 0), (d, 1)); the graph convolution of models/stgcn.py:45-55 in its general form)."""

    @staticmethod
    def forward(ctx, cfg, x, w, b):
        k, s, d, pad = cfg
        x = x.contiguous()
        y, _ = ops.conv(S(x), K=x.shape[1], w=w, bias=b, M=w.shape[0], KT=k, dil=d, stride=s, pad=pad)
        ctx.cfg = cfg
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        k, s, d, pad = ctx.cfg
        dy = dy.contiguous()
        M, K, T = w.shape[0], x.shape[1], x.shape[2]
        dw = ops.wgrad(S(dy), S(x), M=M, K=K, KT=k, dil=d, stride=s, pad=pad).reshape(w.shape)
        db = dy.sum((0, 2, 3))
        dx = None
        if ctx.needs_input_grad[1]:
            if k == 1:
                dx = ops.zeros_like(x) if (s > 1 or pad > 0) else ops.empty_like(x)
                if pad > 0:
                    raise RuntimeError('tam_gcn_amd: a 1 x 1 convolution with temporal padding is not built')
                ops.conv(S(dy), K=M, w=w.reshape(M, K), bias=None, M=K, wmode=1, y=dx, T_out=dy.shape[2], ostride=s)
            else:
                dx, _ = ops.conv(S(dy), K=M, w=w, bias=None, M=K, KT=k, dil=d, stride=1, pad=(k - 1) * d - pad, wmode=1, up=s, T_out=T)
        return None, dx, dw, db


class PointwiseConvFn(_Fn):
    """1x1 convolution with bias over (N, C, T, V) (ST-GCN's `fcn` applied per position in extract_feature, stgcn.py:218-219)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous()
        y, _ = ops.conv(S(x), K=x.shape[1], w=w.reshape(w.shape[0], -1), bias=b, M=w.shape[0])
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        M, K = w.shape[0], x.shape[1]
        dw = ops.wgrad(S(dy), S(x), M=M, K=K).reshape(w.shape)
        db = dy.sum((0, 2, 3))
        dx = None
        if ctx.needs_input_grad[0]:
            dx, _ = ops.conv(S(dy), K=M, w=w.reshape(M, K), bias=None, M=K, wmode=1)
        return dx, dw, db


from . import torch_ops  # noqa: E402,F401  (registers torch.ops.tamgcn.*; imports only .ops)
