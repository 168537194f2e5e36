"""Small-batch eval-mode engine: `Model.forward` for the inference callers of the reference --
ensemble/ensemble_ctrgcn_resnet_eval.py:147-183, models/resnet_gcn_attention.py:82-85 (frozen backbone), visual.py:53-55 --
which push 1..16 clips at a time through a model in eval() mode.

Every TCN_GCN_unit (reference models/ctrgcn.py:266-284) runs as FIVE launches of the f2 kernel family (csrc/f2.hip:
tamgcn_f2_e / _f2_gcn / _f2_gemm x 2 / _f2_tcn) instead of the ~12 training-size launches of the launch-fused eval path: 53
launches per forward instead of 118 (+ 98 concatenation / copy kernels when the model is not in a ParamArena), each sized for
one clip (50..130 workgroups, operands staged once).  Eval-mode BatchNorm is a per-channel affine of the running statistics;
it is folded into the neighbouring weights HERE, once per parameter state (a cheap version check per call re-folds after
an optimiser step, load_state_dict or a train-mode forward).

    eng = FusedEval(model)            # model.eval(); V = 20 joints, the block plan of models.ctrgcn.Model
    logits = eng(x)                   # x (N, C, T, V, M) on the GPU, under torch.no_grad()

`Model.forward` routes here by itself in eval mode without autograd for batches of at most F2_MAX_CLIPS clip-persons
(TAMGCN_F2=0 switches the routing off); `inference.GraphedForward` captures whichever path `Model.forward` takes.
There is no CPU path and no fallback inside: unsupported geometry raises `Unsupported` BEFORE anything is launched and the
caller (Model.forward) then takes the general eval path."""
import ctypes as C
import os
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from . import functional as Fn

__all__ = ['FusedEval', 'Unsupported', 'F2_MAX_CLIPS', 'enabled']

F2_MAX_CLIPS = int(os.environ.get('TAMGCN_F2_MAX_CLIPS', '32'))      # clip-persons (N*M) up to which Model.forward routes here


def enabled():
    return os.environ.get('TAMGCN_F2', '1') != '0'


class Unsupported(RuntimeError):
    """The model / input is outside what the f2 kernels are built for (V = 20, three subsets, k = 1 residual convs, ...)."""


def _affine(bn):
    """eval-mode BatchNorm (functional.BN view) as y = s*x + t"""
    s = bn.w.detach() * torch.rsqrt(bn.rv + bn.eps)
    return s, bn.b.detach() - bn.rm * s


def _fold(w2d, bias, bn):
    s, t = _affine(bn)
    return (w2d.detach() * s[:, None]).contiguous(), (bias.detach() * s + t).contiguous()


class _Block:
    """Folded tensors and geometry of one TCN_GCN_unit."""

    def __init__(self, blk, device):
        g = blk.gcn1
        P = g._pack(g._tensors(device))
        tt = blk.tcn1._tensors()
        if blk._rmode == 'conv':
            r = blk.residual
            tt = tt + [r.conv.weight, r.conv.bias, r.bn.weight, r.bn.bias]
        Q = blk._pack_tcn(tt)
        if P.S != 3:
            raise Unsupported(f'{P.S} subsets (the f2 kernels are built for 3)')
        if P.R > 32 or P.Cin > 256 or P.Cout % 16:
            raise Unsupported(f'unit_gcn({P.Cin}, {P.Cout}) with {P.R} relation channels')
        if Q.Cb % 16 or Q.Cb > 64 or (Q.nb + 2) * Q.Cb != Q.Cout or Q.nb > 4 or len(set(Q.ks)) != 1:
            raise Unsupported(f'MultiScale_TemporalConv with {Q.nb} temporal branches of {Q.Cb} channels, kernels {Q.ks}')
        if Q.rmode == 'conv' and Q.rk != 1:
            raise Unsupported('residual unit_tcn with kernel_size != 1')
        self.Cin, self.Cout, self.R, self.S = P.Cin, P.Cout, P.R, P.S
        self.gmode = {'zero': 0, 'identity': 1, 'conv': 2}[P.mode]
        self.W12, self.B12 = P.W12.detach().contiguous(), P.B12.detach().contiguous()
        self.W3, self.B3 = P.W3.detach().contiguous(), P.B3.detach().contiguous()
        self.W4, self.B4 = P.W4.detach().contiguous(), P.B4.detach().contiguous()
        self.PA, self.alpha = P.PA.detach().float().contiguous(), P.alpha.detach()
        self.sy, self.ty = (t.contiguous() for t in _affine(P.bn))
        self.Wd = self.bd = None
        if P.mode == 'conv':
            self.Wd, self.bd = _fold(P.Wd.reshape(P.Cout, P.Cin), P.bd, P.bnd)
        self.Wo, self.bo = _fold(P.Wo.reshape(P.Cout, P.Cout), P.bo, P.bno)
        # MS-TCN: entry convs of the temporal and pooled branches, then the plain branch: one (Cout x Cout) product
        Cb, nb = Q.Cb, Q.nb
        s_in = torch.cat([_affine(b)[0] for b in Q.bn_in]); t_in = torch.cat([_affine(b)[1] for b in Q.bn_in])
        win = Q.Win.detach().reshape((nb + 1) * Cb, Q.Cin) * s_in[:, None]
        bin_ = Q.bin.detach() * s_in + t_in
        wl, bl = _fold(Q.Wl.reshape(Cb, Q.Cin), Q.bl, Q.bn_l)
        self.We, self.be = torch.cat((win, wl)).contiguous(), torch.cat((bin_, bl)).contiguous()
        self.Cb, self.nb, self.ks, self.dils, self.stride = Cb, nb, int(Q.ks[0]), [int(d) for d in Q.dils], int(Q.stride)
        self.Wt, self.bt = [], []
        for b in range(nb):
            w, bb = _fold(Q.Wt[b].reshape(Cb, Cb * self.ks), Q.bt[b], Q.bn_t[b])
            self.Wt.append(w); self.bt.append(bb)
        self.sp, self.tp = (t.contiguous() for t in _affine(Q.bn_pool))
        self.rmode = {'zero': 0, 'identity': 1, 'conv': 2}[Q.rmode]
        self.Wr = self.br = None
        if Q.rmode == 'conv':
            self.Wr, self.br = _fold(Q.Wr.reshape(Q.Cout, -1), Q.br, Q.bnr)
        # the registered op's arguments: tensors in PARAMS order (an absent one is an empty tensor), geometry as integers
        none = self.sy.new_empty(0)
        self.params = [self.W12, self.B12, self.W3, self.B3, self.W4, self.B4, self.PA, self.alpha, self.sy, self.ty,
                       none if self.Wd is None else self.Wd, none if self.bd is None else self.bd, self.Wo, self.bo, self.We, self.be,
                       self.sp, self.tp, none if self.Wr is None else self.Wr, none if self.br is None else self.br]
        for w, bb in zip(self.Wt, self.bt):
            self.params += [w, bb]
        self.geom = [self.R, self.gmode, self.Cb, self.nb, self.ks, self.stride, self.rmode] + self.dils


def _ptr(t):
    return None if t is None else t.data_ptr()


class FusedEval:
    def __init__(self, model):
        if model.training:
            raise ValueError('FusedEval: put the model in eval() mode first')
        if getattr(model, 'num_point', None) != 20:
            raise Unsupported(f'{getattr(model, "num_point", None)} joints (the f2 kernels are built for V = 20)')
        self.model = model
        self.lib = _lib.load()
        self._watch = list(model.parameters()) + list(model.buffers())
        # train-mode forwards rewrite the running statistics through raw pointers (Tensor._version does not see that):
        # the per-BatchNorm update counters of functional.py do
        self._epochs = [Fn._bn_epoch(m) for m in model.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
        self._key = None
        self._blocks = None
        self._mods = None

    # ---- folded parameters, re-derived when any parameter or buffer changed ------------------------------------------
    def _state_key(self):
        m = self.model
        mods = tuple(id(sub_) for sub_ in m.modules())            # any replaced sub-module (model.l5.gcn1 = ...) re-collects
        if mods != self._mods:                                     # a block was replaced: re-collect what to watch
            self._mods = mods
            self._watch = list(m.parameters()) + list(m.buffers())
            self._epochs = [Fn._bn_epoch(b) for b in m.modules() if isinstance(b, torch.nn.modules.batchnorm._BatchNorm)]
        # parameters inside a ParamArena are views of its flat buffer: an update of that buffer (flat SGD, a broadcast) does not
        # bump their _version, the arena's own state_version() does
        arenas = {}
        for t in self._watch:
            a = getattr(t, '_tamgcn_arena', None)
            if a is not None:
                arenas[id(a)] = a
        return (tuple(t._version for t in self._watch) + tuple(e[0] for e in self._epochs) +
                tuple(v for a in arenas.values() for v in a.state_version()) + (self._watch[0].data_ptr(), mods))

    def _packed(self, device):
        key = self._state_key()
        if self._blocks is None or key != self._key:
            with torch.no_grad():
                self._blocks = [_Block(getattr(self.model, f'l{i}'), device) for i in range(1, 11)]
            self._key = self._state_key()
        return self._blocks

    # ---- one block ---------------------------------------------------------------------------------------------------
    def _block(self, b, x, st=None, xpart=None, want_xpart=False):
        out, xp = torch.ops.tamgcn.tcn_gcn_unit_eval(x, xpart, b.params, b.geom)
        return (out, xp) if want_xpart else out

    # ---- the model ---------------------------------------------------------------------------------------------------
    def blocks(self, x):
        """(N, C, T, V, M) or (N, T, V*C) -> (N*M, 256, T/4, V), N, M   (reference models/ctrgcn.py:324-342)"""
        m = self.model
        if torch.is_grad_enabled() and any(p.requires_grad for p in m.parameters()):
            raise RuntimeError('FusedEval is an inference path: call it under torch.no_grad()')
        if m.training:
            raise RuntimeError('FusedEval: the model went back to train() mode')
        if not x.is_cuda or x.dtype != torch.float32:
            raise RuntimeError('FusedEval: expected a float32 HIP (cuda) tensor; there is no CPU path')
        if x.dim() == 3:
            N, T, VC = x.shape
            x = x.view(N, T, m.num_point, -1).permute(0, 3, 1, 2).contiguous().unsqueeze(-1)
        N, C_, T, V, M = x.shape
        if V != 20:
            raise Unsupported(f'{V} joints')
        blocks = self._packed(x.device)
        h = Fn.StemFn.run(m.data_bn, x.contiguous(), m.data_bn.weight, m.data_bn.bias)
        xp = None
        for i, b in enumerate(blocks):
            h, xp = self._block(b, h, xpart=xp, want_xpart=True)
        return h, N, M

    def __call__(self, x):
        h, N, M = self.blocks(x)
        m = self.model
        if isinstance(m.drop_out, torch.nn.Dropout):           # eval mode: dropout is the identity
            pass
        return torch.ops.tamgcn.head(h, m.fc.weight, m.fc.bias, M)

    forward = __call__


# ----------------------------------------------------------------------------------------------------------------------
# The block as a registered operator (BASELINE.json north_star: "registering custom ops through a thin C-ABI"): pure
# tensors in, pure tensors out, a fake implementation for tracing / export.
#   params: W12 [3*2R][Cin], B12, W3 [3*Cout][Cin], B3, W4 [3][Cout][R], B4, PA [3][V][V], alpha [1], sy, ty [Cout] (unit_gcn.bn
#           folded), Wd [Cout][Cin], bd (down, folded; empty if none), Wo [Cout][Cout], bo (offset_conv folded), We [Cout][Cout], be
#           (entry convs of the temporal + pooled branches, then the plain branch, folded), sp, tp [Cb] (pooled branch's
#           BatchNorm), Wr [Cout][Cres], br (residual unit_tcn folded; empty if none), then Wt_b [Cb][Cb*ks], bt_b per branch
#   geom:   R, unit_gcn residual (0 zero | 1 identity | 2 conv), Cb, nb, ks, stride, block residual (0 | 1 | 2), dilations
# Returns (out (N, Cout, T2, V), xpart (N, ceil(T2/4), Cout, V)): xpart = per-tile frame sums of out, the next block's input.
# ----------------------------------------------------------------------------------------------------------------------
def _opt(t):
    return None if t is None or t.numel() == 0 else t.data_ptr()


@torch.library.custom_op('tamgcn::tcn_gcn_unit_eval', mutates_args=())
def tcn_gcn_unit_eval(x: Tensor, xpart: Optional[Tensor], params: List[Tensor], geom: List[int]) -> Tuple[Tensor, Tensor]:
    lib = _lib.load()
    if not x.is_cuda or x.dtype != torch.float32:
        raise RuntimeError('tamgcn::tcn_gcn_unit_eval: expected a float32 HIP (cuda) tensor; there is no CPU path')
    x = x.contiguous()
    (W12, B12, W3, B3, W4, B4, PA, alpha, sy, ty, Wd, bd, Wo, bo, We, be, sp, tp, Wr, br), rest = params[:20], params[20:]
    R, gmode, Cb, nb, ks, stride, rmode = geom[:7]
    dils = geom[7:7 + nb]
    N, Cin, T, V = x.shape
    Cout = W3.shape[0] // 3
    dev = x.device
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    E = torch.empty(N, 3, Cout, V, V, device=dev)
    ws = torch.empty(4, N, Cout, T, V, device=dev)               # y + res, res - y, g, h
    sm, df, g, h = ws[0], ws[1], ws[2], ws[3]
    d = _lib.F2GcnDesc(N=N, Cin=Cin, Cout=Cout, T=T, V=V, S=3, R=R, res_mode=gmode,
                       x=x.data_ptr(), w12=W12.data_ptr(), b12=B12.data_ptr(), w4=W4.data_ptr(), b4=B4.data_ptr(),
                       A=PA.data_ptr(), alpha=alpha.data_ptr(), w3=W3.data_ptr(), b3=B3.data_ptr(),
                       sy=sy.data_ptr(), ty=ty.data_ptr(), wd=_opt(Wd), bd=_opt(bd),
                       E=E.data_ptr(), sum=sm.data_ptr(), diff=df.data_ptr(), xpart=_opt(xpart))
    _lib.check(lib.tamgcn_f2_e(C.byref(d), st), 'tamgcn_f2_e')
    _lib.check(lib.tamgcn_f2_gcn(C.byref(d), st), 'tamgcn_f2_gcn')
    q = _lib.F2GemmDesc(N=N, K=Cout, M=Cout, T=T, V=V, mode=0, relu_rows=0, x=df.data_ptr(), w=Wo.data_ptr(), b=bo.data_ptr(),
                        add=sm.data_ptr(), out=g.data_ptr())
    _lib.check(lib.tamgcn_f2_gemm(C.byref(q), st), 'tamgcn_f2_gemm')
    q = _lib.F2GemmDesc(N=N, K=Cout, M=Cout, T=T, V=V, mode=1, relu_rows=(nb + 1) * Cb, x=g.data_ptr(), w=We.data_ptr(),
                        b=be.data_ptr(), add=None, out=h.data_ptr())
    _lib.check(lib.tamgcn_f2_gemm(C.byref(q), st), 'tamgcn_f2_gemm')
    T2 = (T - 1) // stride + 1
    out = torch.empty(N, Cout, T2, V, device=dev)
    xp = torch.empty(N, (T2 + 3) // 4, Cout, V, device=dev)      # per-tile frame sums: the next block's xbar
    t = _lib.F2TcnDesc(N=N, Cin=Wr.shape[1] if rmode == 2 else Cin, Cout=Cout, T=T, V=V, stride=stride, Cb=Cb, nb=nb, ks=ks,
                       res_mode=rmode, h=h.data_ptr(), sp=sp.data_ptr(), tp=tp.data_ptr(), x=x.data_ptr(), wr=_opt(Wr), br=_opt(br),
                       out=out.data_ptr(), xpart=xp.data_ptr())
    for i in range(nb):
        t.dil[i] = dils[i]
        t.wt[i] = rest[2 * i].data_ptr()
        t.bt[i] = rest[2 * i + 1].data_ptr()
    _lib.check(lib.tamgcn_f2_tcn(C.byref(t), st), 'tamgcn_f2_tcn')
    return out, xp


@tcn_gcn_unit_eval.register_fake
def _(x, xpart, params, geom):
    N, _, T, V = x.shape
    Cout = params[2].shape[0] // 3
    T2 = (T - 1) // geom[5] + 1
    return x.new_empty(N, Cout, T2, V), x.new_empty(N, (T2 + 3) // 4, Cout, V)
