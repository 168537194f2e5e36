"""Drop-in mirror of the reference's ``models.ctrgcn`` module surface
(/root/reference/models/ctrgcn.py) on top of the MI355X-native HIP ops.

Same public names, constructor signatures, parameter/buffer names (the
state-dict is the checkpoint ABI: 892 keys for the N-UCLA model),
initial-value distributions and RNG consumption order (so that
``torch.manual_seed(s); Model(...)`` gives the same initial state-dict as the
reference).  The nn.Conv2d / nn.BatchNorm2d / nn.Sequential children are used
as *parameter containers only*: no forward here ever calls them.  Every forward
sequences hand-written gfx950 kernels through ``tam_gcn_amd.functional``; a
CPU tensor raises (there is no CPU fallback).

Reference lines: TemporalConv :52-69, MultiScale_TemporalConv :72-147,
CTRGC :150-177, unit_tcn :179-193, unit_gcn :196-263, TCN_GCN_unit :266-284,
Model :287-375, init helpers :17-49.
"""
import importlib
import math

import numpy as np
import torch
import torch.nn as nn

from .. import functional as Fn
from .. import torch_ops  # noqa: F401  (registers torch.ops.tamgcn.*)


def import_class(name):
    """Resolve a dotted class path (reference :9-14).  ``graph.ucla.Graph`` style
    paths resolve against an importable top-level ``graph`` package when there is
    one (running inside the reference checkout), else against ``tam_gcn_amd.graph``."""
    parts = name.split('.')
    for root in (parts[0], 'tam_gcn_amd.' + parts[0]):
        try:
            mod = importlib.import_module(root)
            for comp in parts[1:-1]:
                mod = importlib.import_module(mod.__name__ + '.' + comp)
            return getattr(mod, parts[-1])
        except (ImportError, AttributeError):
            continue
    raise ImportError(f'cannot resolve {name}')


# ---------------------------------------------------------------------------
# init helpers (reference :17-49)
# ---------------------------------------------------------------------------
def conv_branch_init(conv, branches):
    w = conv.weight
    nn.init.normal_(w, 0, math.sqrt(2. / (w.size(0) * w.size(1) * w.size(2) * branches)))
    nn.init.constant_(conv.bias, 0)


def conv_init(conv):
    if conv.weight is not None:
        nn.init.kaiming_normal_(conv.weight, mode='fan_out')
    if conv.bias is not None:
        nn.init.constant_(conv.bias, 0)


def bn_init(bn, scale):
    nn.init.constant_(bn.weight, scale)
    nn.init.constant_(bn.bias, 0)


def weights_init(m):
    # the reference dispatches on the class *name* containing 'Conv' / 'BatchNorm';
    # for the module types that occur here that is equivalent to:
    if isinstance(m, nn.Conv2d):
        nn.init.kaiming_normal_(m.weight, mode='fan_out')
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.modules.batchnorm._BatchNorm):
        if m.weight is not None:
            m.weight.data.normal_(1.0, 0.02)
        if m.bias is not None:
            m.bias.data.fill_(0)


def _w2(conv):
    """(O, I, k, 1) conv weight as a contiguous (O, I*k) matrix view."""
    w = conv if isinstance(conv, torch.Tensor) else conv.weight
    return w.reshape(w.shape[0], -1)


def _cat(ts, stack=False):
    """torch.cat / torch.stack of the tensors -- without a copy when they already sit back to back in memory, which is
    how tam_gcn_amd.distributed.ParamArena lays the packed groups out (60 concatenation kernels per step otherwise)."""
    t0 = ts[0]
    ptr, ok = t0.data_ptr(), True
    base = t0.untyped_storage().data_ptr()             # back to back is not enough: they must live in ONE storage
    for t in ts:
        if not t.is_contiguous() or t.data_ptr() != ptr or t.dtype != t0.dtype or t.untyped_storage().data_ptr() != base:
            ok = False
            break
        ptr += t.numel() * t.element_size()
    if not ok:
        return torch.stack(ts) if stack else torch.cat(ts)
    shape = (len(ts),) + tuple(t0.shape) if stack else (sum(t.shape[0] for t in ts),) + tuple(t0.shape[1:])
    n = 1
    for d in shape:
        n *= d
    return torch.as_strided(t0, (n,), (1,)).view(shape)      # same storage: the arena


def _require_hip(x):
    if not x.is_cuda:
        raise RuntimeError('tam_gcn_amd: the CTR-GCN hot path runs on MI355X only (got a CPU tensor); '
                           'there is no CPU fallback')
    if x.dtype != torch.float32:
        raise RuntimeError(f'tam_gcn_amd: fp32 activations expected, got {x.dtype}')
    return x.contiguous()


# ---------------------------------------------------------------------------
class TemporalConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dilation=1):
        super().__init__()
        self.kernel_size, self.stride, self.dilation = kernel_size, stride, dilation
        self.pad = (kernel_size + (kernel_size - 1) * (dilation - 1) - 1) // 2
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=(kernel_size, 1), padding=(self.pad, 0),
                              stride=(stride, 1), dilation=(dilation, 1))
        self.bn = nn.BatchNorm2d(out_channels)
        Fn.tag_batchnorms_(self)

    def forward(self, x):
        cfg = (self.kernel_size, self.stride, self.dilation, self.pad, self.bn)
        return Fn.ConvBNFn.run(cfg, _require_hip(x), self.conv.weight, self.conv.bias, self.bn.weight, self.bn.bias)


class unit_tcn(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=9, stride=1):
        super().__init__()
        self.kernel_size, self.stride = kernel_size, stride
        self.pad = int((kernel_size - 1) / 2)
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=(kernel_size, 1), padding=(self.pad, 0),
                              stride=(stride, 1))
        self.bn = nn.BatchNorm2d(out_channels)
        self.relu = nn.ReLU(inplace=True)          # kept for parity; never applied (reference :191-193)
        conv_init(self.conv)
        bn_init(self.bn, 1)
        Fn.tag_batchnorms_(self)

    def forward(self, x):
        cfg = (self.kernel_size, self.stride, 1, self.pad, self.bn)
        return Fn.ConvBNFn.run(cfg, _require_hip(x), self.conv.weight, self.conv.bias, self.bn.weight, self.bn.bias)


# ---------------------------------------------------------------------------
class MultiScale_TemporalConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, dilations=[1, 2, 3, 4],
                 residual=True, residual_kernel_size=1):
        super().__init__()
        assert out_channels % (len(dilations) + 2) == 0, '# out channels should be multiples of # branches'
        self.num_branches = len(dilations) + 2
        bc = out_channels // self.num_branches
        if type(kernel_size) == list:
            assert len(kernel_size) == len(dilations)
        else:
            kernel_size = [kernel_size] * len(dilations)
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride
        self._ks, self._dils, self._bc = list(kernel_size), list(dilations), bc

        def entry(**kw):
            return [nn.Conv2d(in_channels, bc, kernel_size=1, padding=0, **kw), nn.BatchNorm2d(bc)]

        self.branches = nn.ModuleList(
            nn.Sequential(*entry(), nn.ReLU(inplace=True), TemporalConv(bc, bc, kernel_size=k, stride=stride, dilation=d))
            for k, d in zip(kernel_size, dilations))
        self.branches.append(nn.Sequential(*entry(), nn.ReLU(inplace=True),
                                           nn.MaxPool2d(kernel_size=(3, 1), stride=(stride, 1), padding=(1, 0)),
                                           nn.BatchNorm2d(bc)))
        self.branches.append(nn.Sequential(*entry(stride=(stride, 1))))
        if not residual:
            self._rmode = 'zero'
            self.residual = lambda x: 0
        elif in_channels == out_channels and stride == 1:
            self._rmode = 'identity'
            self.residual = lambda x: x
        else:
            self._rmode = 'conv'
            self.residual = TemporalConv(in_channels, out_channels, kernel_size=residual_kernel_size, stride=stride)
        self._rk = residual_kernel_size
        self.apply(weights_init)
        Fn.tag_batchnorms_(self)

    # ---- parameter plumbing -------------------------------------------------
    def _tensors(self):
        nb = len(self._dils)
        t = []
        for b in range(nb):
            br = self.branches[b]
            t += [br[0].weight, br[0].bias, br[1].weight, br[1].bias,
                  br[3].conv.weight, br[3].conv.bias, br[3].bn.weight, br[3].bn.bias]
        br = self.branches[nb]
        t += [br[0].weight, br[0].bias, br[1].weight, br[1].bias, br[4].weight, br[4].bias]
        br = self.branches[nb + 1]
        t += [br[0].weight, br[0].bias, br[1].weight, br[1].bias]
        if self._rmode == 'conv':
            r = self.residual
            t += [r.conv.weight, r.conv.bias, r.bn.weight, r.bn.bias]
        return t

    def _arena_groups(self):
        """Parameter lists _pack() concatenates, in that order (ParamArena keeps each back to back)."""
        nb = len(self._dils)
        heads = [self.branches[b][0] for b in range(nb + 1)]
        return [[h.weight for h in heads], [h.bias for h in heads]]

    def _pack(self, params, ext_res=None, relu=False):
        """params in _tensors() order (+ 4 tensors of an external unit_tcn residual)."""
        nb = len(self._dils)
        P = Fn.TcnParams()
        P.Cin, P.Cout, P.Cb, P.nb = self.in_channels, self.out_channels, self._bc, nb
        P.ks, P.dils, P.stride, P.relu = self._ks, self._dils, self.stride, relu
        it = iter(params)
        wins, bins = [], []
        P.bn_in, P.Wt, P.bt, P.bn_t = [], [], [], []
        for b in range(nb):
            w, bia, _, _, wt, bt, _, _ = (next(it) for _ in range(8))
            wins.append(_w2(w)); bins.append(bia)
            P.Wt.append(wt); P.bt.append(bt)
            P.bn_in.append(Fn.BN(self.branches[b][1])); P.bn_t.append(Fn.BN(self.branches[b][3].bn))
        w, bia, _, _, _, _ = (next(it) for _ in range(6))
        wins.append(_w2(w)); bins.append(bia)
        P.bn_in.append(Fn.BN(self.branches[nb][1])); P.bn_pool = Fn.BN(self.branches[nb][4])
        w, bia, _, _ = (next(it) for _ in range(4))
        P.Wl, P.bl, P.bn_l = w, bia, Fn.BN(self.branches[nb + 1][1])
        P.Win, P.bin = _cat(wins), _cat(bins)
        P.Wr = P.br = P.bnr = None
        P.rk = 1
        if ext_res is not None:                       # TCN_GCN_unit's own residual
            P.rmode = ext_res[0]
            if P.rmode == 'conv':
                w, bia, _, _ = (next(it) for _ in range(4))
                P.Wr, P.br, P.bnr, P.rk = w, bia, Fn.BN(ext_res[1].bn), ext_res[1].kernel_size
        else:
            P.rmode = self._rmode
            if P.rmode == 'conv':
                w, bia, _, _ = (next(it) for _ in range(4))
                P.Wr, P.br, P.bnr, P.rk = w, bia, Fn.BN(self.residual.bn), self._rk
        return P

    def _route(self, G, ext_conv=False):
        nb = len(self._dils)
        Cb, Cin = self._bc, self.in_channels
        out = []
        dWin, dbin = G['Win'], G['bin']                # lists: one tensor per branch
        for b in range(nb):
            out += [dWin[b], dbin[b], G['bn_in'][b][0], G['bn_in'][b][1], G['Wt'][b], G['bt'][b], G['bn_t'][b][0], G['bn_t'][b][1]]
        out += [dWin[nb], dbin[nb], G['bn_in'][nb][0], G['bn_in'][nb][1], G['bn_pool'][0], G['bn_pool'][1]]
        out += [G['Wl'], G['bl'], G['bn_l'][0], G['bn_l'][1]]
        if self._rmode == 'conv' or ext_conv:
            out += [G['Wr'], G['br'], G['bnr'][0], G['bnr'][1]]
        return out

    def forward(self, x):
        return Fn.MSTCNFn.run(self, _require_hip(x), *self._tensors())


# ---------------------------------------------------------------------------
class CTRGC(nn.Module):
    def __init__(self, in_channels, out_channels, rel_reduction=8, mid_reduction=1):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        if in_channels == 3 or in_channels == 9:
            self.rel_channels, self.mid_channels = 8, 16
        else:
            self.rel_channels = in_channels // rel_reduction
            self.mid_channels = in_channels // mid_reduction
        # the refinement kernels (csrc/ctrgc.hip, ctrgc_de.hip) are built for R = 4, 8, ..., 32 rel-channels: every width the
        # reference's models use (in_channels 3 / 9 -> 8; 64 / 128 / 256 -> 8 / 16 / 32).  Say so HERE, not at the first forward
        if self.rel_channels < 4 or self.rel_channels > 32 or self.rel_channels % 4:
            raise NotImplementedError(
                f'tam_gcn_amd CTRGC: rel_channels = in_channels // rel_reduction = {self.rel_channels} is not built '
                f'(in_channels {in_channels}, rel_reduction {rel_reduction}); supported: multiples of 4 up to 32, i.e. '
                f'in_channels // rel_reduction in {{4, 8, ..., 32}} (in_channels 3 and 9 use 8, as in the reference)')
        self.conv1 = nn.Conv2d(in_channels, self.rel_channels, kernel_size=1)
        self.conv2 = nn.Conv2d(in_channels, self.rel_channels, kernel_size=1)
        self.conv3 = nn.Conv2d(in_channels, out_channels, kernel_size=1)
        self.conv4 = nn.Conv2d(self.rel_channels, out_channels, kernel_size=1)
        self.tanh = nn.Tanh()
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                conv_init(m)
            elif isinstance(m, nn.BatchNorm2d):
                bn_init(m, 1)

    def _tensors(self):
        return [self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias,
                self.conv3.weight, self.conv3.bias, self.conv4.weight, self.conv4.bias]

    def forward(self, x, A=None, alpha=1):
        x = _require_hip(x)
        V = x.shape[-1]
        if A is None:
            A = torch.zeros(V, V, device=x.device)
        if not isinstance(alpha, torch.Tensor):
            alpha = torch.tensor([float(alpha)], device=x.device)
        return torch.ops.tamgcn.ctrgc(x, A, alpha, *self._tensors())


# ---------------------------------------------------------------------------
class unit_gcn(nn.Module):
    def __init__(self, in_channels, out_channels, A, coff_embedding=4, adaptive=True, residual=True):
        super().__init__()
        self.inter_c = out_channels // coff_embedding
        self.out_c, self.in_c = out_channels, in_channels
        self.adaptive = adaptive
        self.num_subset = A.shape[0]
        self.convs = nn.ModuleList(CTRGC(in_channels, out_channels) for _ in range(self.num_subset))
        if residual:
            if in_channels != out_channels:
                self._mode = 'conv'
                self.down = nn.Sequential(nn.Conv2d(in_channels, out_channels, 1), nn.BatchNorm2d(out_channels))
            else:
                self._mode = 'identity'
                self.down = lambda x: x
        else:
            self._mode = 'zero'
            self.down = lambda x: 0
        self.offset_conv = nn.Sequential(nn.Conv2d(out_channels, out_channels, 1), nn.BatchNorm2d(out_channels),
                                         nn.Tanh())
        if adaptive:
            self.PA = nn.Parameter(torch.from_numpy(A.astype(np.float32)))
        else:
            self.A = torch.from_numpy(A.astype(np.float32))
        self.alpha = nn.Parameter(torch.zeros(1))
        self.bn = nn.BatchNorm2d(out_channels)
        self.soft = nn.Softmax(-2)                 # dead members kept for API parity (reference :231-232)
        self.relu = nn.ReLU(inplace=True)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                conv_init(m)
            elif isinstance(m, nn.BatchNorm2d):
                bn_init(m, 1)
        bn_init(self.bn, 1e-6)
        for m in self.offset_conv.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.constant_(m.weight, 0)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        Fn.tag_batchnorms_(self)

    # ---- parameter plumbing -------------------------------------------------
    def _graph(self, device):
        if self.adaptive:
            return self.PA
        if self.A.device != device:
            self.A = self.A.to(device)
        return self.A

    def _tensors(self, device=None):
        t = [self._graph(device), self.alpha]
        for c in self.convs:
            t += c._tensors()
        t += [self.bn.weight, self.bn.bias]
        if self._mode == 'conv':
            t += [self.down[0].weight, self.down[0].bias, self.down[1].weight, self.down[1].bias]
        t += [self.offset_conv[0].weight, self.offset_conv[0].bias, self.offset_conv[1].weight, self.offset_conv[1].bias]
        return t

    def _arena_groups(self):
        """Parameter lists _pack() concatenates, in that order (ParamArena keeps each back to back)."""
        cs = self.convs
        return [[w for c in cs for w in (c.conv1.weight, c.conv2.weight)], [b for c in cs for b in (c.conv1.bias, c.conv2.bias)],
                [c.conv3.weight for c in cs], [c.conv3.bias for c in cs], [c.conv4.weight for c in cs], [c.conv4.bias for c in cs]]

    def _pack(self, params):
        P = Fn.GcnParams()
        S_, Cin, Cout = self.num_subset, self.in_c, self.out_c
        R = self.convs[0].rel_channels
        P.S, P.R, P.Cin, P.Cout, P.mode = S_, R, Cin, Cout, self._mode
        P.PA, P.alpha = params[0].contiguous(), params[1]
        w12, b12, w3, b3, w4, b4 = [], [], [], [], [], []
        for i in range(S_):
            w1, b1, w2, b2, w3_, b3_, w4_, b4_ = params[2 + 8 * i: 10 + 8 * i]
            w12 += [_w2(w1), _w2(w2)]; b12 += [b1, b2]
            w3.append(_w2(w3_)); b3.append(b3_); w4.append(_w2(w4_)); b4.append(b4_)
        P.W12, P.B12 = _cat(w12), _cat(b12)
        P.W3, P.B3 = _cat(w3), _cat(b3)
        P.W4, P.B4 = _cat(w4, stack=True), _cat(b4, stack=True)
        k = 2 + 8 * S_
        P.bn = Fn.BN(self.bn)
        k += 2
        P.Wd = P.bd = P.bnd = None
        if self._mode == 'conv':
            P.Wd, P.bd = params[k], params[k + 1]
            P.bnd = Fn.BN(self.down[1])
            k += 4
        P.Wo, P.bo = params[k], params[k + 1]
        P.bno = Fn.BN(self.offset_conv[1])
        return P

    def _route(self, G):
        S_, Cin, Cout = self.num_subset, self.in_c, self.out_c
        R = self.convs[0].rel_channels
        out = [G['PA'], G['alpha']]
        W12, W3, B3, W4, B4 = G['W12'], G['W3'], G['B3'], G['W4'], G['B4']     # lists: one tensor per parameter
        B12 = G['B12'].reshape(S_, 2, R)
        for i in range(S_):
            out += [W12[2 * i], B12[i, 0], W12[2 * i + 1], B12[i, 1], W3[i], B3[i], W4[i], B4[i]]
        out += [G['bn.w'], G['bn.b']]
        if self._mode == 'conv':
            out += [G['Wd'].reshape(Cout, Cin, 1, 1), G['bd'], G['bnd.w'], G['bnd.b']]
        out += [G['Wo'].reshape(Cout, Cout, 1, 1), G['bo'], G['bno.w'], G['bno.b']]
        return out

    def forward(self, x):
        x = _require_hip(x)
        return Fn.UnitGCNFn.run(self, x, *self._tensors(x.device))


# ---------------------------------------------------------------------------
class TCN_GCN_unit(nn.Module):
    def __init__(self, in_channels, out_channels, A, stride=1, residual=True, adaptive=True, kernel_size=5,
                 dilations=[1, 2]):
        super().__init__()
        self.gcn1 = unit_gcn(in_channels, out_channels, A, adaptive=adaptive)
        self.tcn1 = MultiScale_TemporalConv(out_channels, out_channels, kernel_size=kernel_size, stride=stride,
                                            dilations=dilations, residual=False)
        self.relu = nn.ReLU(inplace=True)
        if not residual:
            self._rmode = 'zero'
            self.residual = lambda x: 0
        elif in_channels == out_channels and stride == 1:
            self._rmode = 'identity'
            self.residual = lambda x: x
        else:
            self._rmode = 'conv'
            self.residual = unit_tcn(in_channels, out_channels, kernel_size=1, stride=stride)

    def _pack_tcn(self, params):
        ext = (self._rmode, self.residual if self._rmode == 'conv' else None)
        return self.tcn1._pack(params, ext_res=ext, relu=True)

    def _route_tcn(self, G):
        return self.tcn1._route(G, ext_conv=self._rmode == 'conv')

    def forward(self, x, emit_pool=False):
        """emit_pool (not in the reference's signature; Model uses it for l10): return (out, rowmean) with rowmean
        the (N, C) means over (t, v) of out, taken in the block's last pass -- or an empty tensor if that pass did not
        run (launch-fused eval)."""
        x = _require_hip(x)
        gt = self.gcn1._tensors(x.device)
        tt = self.tcn1._tensors()
        if self._rmode == 'conv':
            r = self.residual
            tt = tt + [r.conv.weight, r.conv.bias, r.bn.weight, r.bn.bias]
        return Fn.TCNGCNUnitFn.run(self, x, len(gt), bool(emit_pool), *gt, *tt)


# ---------------------------------------------------------------------------
class Model(nn.Module):
    def __init__(self, num_class=60, num_point=25, num_person=2, graph=None, graph_args=dict(), in_channels=3,
                 drop_out=0, adaptive=True):
        super().__init__()
        if graph is None:
            raise ValueError()
        Graph = import_class(graph)
        self.graph = Graph(**graph_args)
        A = self.graph.A                                   # (3, V, V) float64
        self.num_class, self.num_point = num_class, num_point
        self.data_bn = nn.BatchNorm1d(num_person * in_channels * num_point)
        c = 64
        plan = [(in_channels, c, 1, False), (c, c, 1, True), (c, c, 1, True), (c, c, 1, True),
                (c, 2 * c, 2, True), (2 * c, 2 * c, 1, True), (2 * c, 2 * c, 1, True),
                (2 * c, 4 * c, 2, True), (4 * c, 4 * c, 1, True), (4 * c, 4 * c, 1, True)]
        for i, (ci, co, s, res) in enumerate(plan, 1):
            setattr(self, f'l{i}', TCN_GCN_unit(ci, co, A, stride=s, residual=res, adaptive=adaptive))
        self.fc = nn.Linear(4 * c, num_class)
        nn.init.normal_(self.fc.weight, 0, math.sqrt(2. / num_class))
        bn_init(self.data_bn, 1)
        self.drop_out = nn.Dropout(drop_out) if drop_out else (lambda x: x)
        Fn.tag_batchnorms_(self)

    def _blocks(self, x, emit_pool=False):
        if x.dim() == 3:                                   # (N, T, V*C) form, reference :325-327
            N, T, VC = x.shape
            x = x.view(N, T, self.num_point, -1).permute(0, 3, 1, 2).contiguous().unsqueeze(-1)
        N, C, T, V, M = x.size()
        # reference :330-332 (permute, BatchNorm1d, permute back) as one statistics pass + one apply-and-permute pass
        x = Fn.StemFn.run(self.data_bn, x, self.data_bn.weight, self.data_bn.bias)
        for i in range(1, 10):
            x = getattr(self, f'l{i}')(x)
        if emit_pool:
            x, rm = self.l10(x, emit_pool=True)
            return x, N, M, rm
        return self.l10(x), N, M

    def _f2(self, x):
        """The small-batch eval engine (tam_gcn_amd.f2) if this call is one for it, else None."""
        if self.training or torch.is_grad_enabled() or not x.is_cuda or x.dtype != torch.float32:
            return None
        from .. import f2
        if not f2.enabled() or (x.shape[0] * (x.shape[4] if x.dim() == 5 else 1)) > f2.F2_MAX_CLIPS:
            return None
        # the engine calls the block operator directly: forward hooks on any sub-module would not fire.  A model that
        # carries hooks (feature extraction, visualisation: visual.py:53-55 style) takes the general path, module by module
        for mod in self.modules():
            if mod._forward_hooks or mod._forward_pre_hooks:
                return None
        eng = self.__dict__.get('_tamgcn_f2')
        if eng and eng.model is not self:                  # an nn.DataParallel replica carries the original's __dict__: its own engine
            eng = None
        if eng is None:
            try:
                eng = f2.FusedEval(self)
                eng._packed(x.device)                      # geometry checks happen here, before anything is launched
            except f2.Unsupported:
                eng = False                                # this model is outside the family: the general eval path serves it
            self.__dict__['_tamgcn_f2'] = eng
        return eng or None

    def forward(self, x):
        x = _require_hip(x)
        eng = self._f2(x)
        if eng is not None:
            return eng(x)
        if isinstance(self.drop_out, nn.Dropout):          # drop_out > 0: pool here, torch's dropout + linear (reference :343-348)
            x, N, M = self._blocks(x)
            x = x.view(N, M, x.size(1), -1).mean(3).mean(1)
            return self.fc(self.drop_out(x))
        x, N, M, rm = self._blocks(x, emit_pool=True)
        if rm.shape[0]:                                    # l10's last pass already reduced its rows: no second pass over x
            return torch.ops.tamgcn.head_pooled(x, rm, self.fc.weight, self.fc.bias, M)
        return torch.ops.tamgcn.head(x, self.fc.weight, self.fc.bias, M)

    def extract_feature(self, x):
        x = _require_hip(x)
        eng = self._f2(x)
        x, N, M = eng.blocks(x) if eng is not None else self._blocks(x)
        _, C, T, V = x.size()
        x = x.view(N, M, C, T, V).permute(0, 2, 3, 4, 1).contiguous()
        return x, x
