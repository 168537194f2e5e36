"""Drop-in mirror of the reference's ``models.stgcn`` module surface (/root/reference/models/stgcn.py) on the HIP ops
(SURVEY.md §8 row f4): same class names, constructor signatures, state-dict keys (``A`` buffer, ``data_bn``,
``st_gcn_networks.{i}.gcn.conv``, ``.tcn.{0,2,3}``, ``.residual.{0,1}``, ``edge_importance.{i}``, ``fcn``) and RNG
consumption order at construction.  The spatial graph convolution reuses the fused CTRGC kernels with a static topology
(functional.StGcnFn); the stem and head are the CTR-GCN model's.  A CPU tensor raises: there is no CPU fallback.

Reference lines: ConvTemporalGraphical :37-64, st_gcn :67-99, Model :102-222, get_edge_importance_per_joint :224-252."""
import numpy as np
import torch
import torch.nn as nn

from .. import functional as Fn
from .ctrgcn import import_class, _require_hip


def conv_init(conv):
    if conv.weight is not None:
        nn.init.kaiming_normal_(conv.weight, mode='fan_out')
    if conv.bias is not None:
        nn.init.constant_(conv.bias, 0)


def bn_init(bn, scale):
    nn.init.constant_(bn.weight, scale)
    nn.init.constant_(bn.bias, 0)


class ConvTemporalGraphical(nn.Module):
    """The spatial graph convolution (reference :37-64).  Inside st_gcn (the 1 x 1 form, the only one the model builds) it
    is a parameter container: the block's fused node computes it on the CTRGC kernels.  Called on its own -- any
    t_kernel_size / t_stride / t_padding / t_dilation, with or without bias -- the convolution runs on the k x 1 HIP
    kernels and the joint contraction einsum('nkctv,kvw->nctw') as the library batched GEMM it is (a plain GEMM with a
    shared (K*V) x V right-hand side: nothing to fuse with, and not on any configuration's path)."""

    def __init__(self, in_channels, out_channels, kernel_size, t_kernel_size=1, t_stride=1, t_padding=0, t_dilation=1, bias=True):
        super().__init__()
        self.kernel_size = kernel_size
        self._cfg = (t_kernel_size, t_stride, t_dilation, t_padding)
        self.conv = nn.Conv2d(in_channels, out_channels * kernel_size, kernel_size=(t_kernel_size, 1), padding=(t_padding, 0),
                              stride=(t_stride, 1), dilation=(t_dilation, 1), bias=bias)

    def forward(self, x, A):
        assert A.size(0) == self.kernel_size
        x = _require_hip(x)
        bias = self.conv.bias if self.conv.bias is not None else torch.zeros(self.conv.out_channels, device=x.device)
        h = Fn.TemporalConvFn.run(self._cfg, x, self.conv.weight, bias)
        n, kc, t, v = h.size()
        h = h.view(n, self.kernel_size, kc // self.kernel_size, t, v)
        return torch.einsum('nkctv,kvw->nctw', (h, A)).contiguous(), A


class st_gcn(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dropout=0, residual=True):
        super().__init__()
        assert len(kernel_size) == 2
        assert kernel_size[0] % 2 == 1
        padding = ((kernel_size[0] - 1) // 2, 0)
        self.in_channels, self.out_channels, self.stride, self.t_kernel = in_channels, out_channels, stride, kernel_size[0]
        if kernel_size[0] not in (1, 3, 5, 9):
            raise NotImplementedError('tam_gcn_amd: temporal kernel sizes 1, 3, 5, 9 are instantiated')
        self.gcn = ConvTemporalGraphical(in_channels, out_channels, kernel_size[1])
        self.tcn = nn.Sequential(
            nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True),
            nn.Conv2d(out_channels, out_channels, (kernel_size[0], 1), (stride, 1), padding),
            nn.BatchNorm2d(out_channels),
            nn.Dropout(dropout, inplace=True),
        )
        self._dropout = dropout
        if not residual:
            self._rmode = 'zero'
            self.residual = lambda x: 0
        elif (in_channels == out_channels) and (stride == 1):
            self._rmode = 'identity'
            self.residual = lambda x: x
        else:
            self._rmode = 'conv'
            self.residual = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=(stride, 1)),
                                          nn.BatchNorm2d(out_channels))
        self.relu = nn.ReLU(inplace=True)
        Fn.tag_batchnorms_(self)

    def forward(self, x, A):
        x = _require_hip(x)
        if A.size(0) != self.gcn.kernel_size:
            raise AssertionError('A.size(0) must equal the spatial kernel size')
        p = [self.gcn.conv.weight, self.gcn.conv.bias, self.tcn[0].weight, self.tcn[0].bias, self.tcn[2].weight, self.tcn[2].bias,
             self.tcn[3].weight, self.tcn[3].bias]
        if self._dropout and self.training:
            # reference :82-88, :96-99 with an active Dropout between the second BatchNorm and the residual add: the fused node
            # stops at bn2(conv(relu(bn1(gcn(x))))); dropout, the residual branch and the final ReLU follow as separate ops
            z = Fn.StGcnFn.run(self, False, x, A, *p)
            z = torch.nn.functional.dropout(z, self._dropout, True)
            if self._rmode == 'identity':
                z = z + x
            elif self._rmode == 'conv':
                r = self.residual
                z = z + Fn.ConvBNFn.run((1, self.stride, 1, 0, r[1]), x, r[0].weight, r[0].bias, r[1].weight, r[1].bias)
            return torch.relu(z), A
        if self._rmode == 'conv':
            p += [self.residual[0].weight, self.residual[0].bias, self.residual[1].weight, self.residual[1].bias]
        return Fn.StGcnFn.run(self, True, x, A, *p), A


class Model(nn.Module):
    def __init__(self, in_channels=3, num_class=4, num_point=20, num_person=1, graph=None, graph_args=dict(),
                 edge_importance_weighting=True, dropout=0, **kwargs):
        super().__init__()
        if graph is None:
            raise ValueError("Graph class must be specified")
        Graph = import_class(graph)
        self.graph = Graph(**graph_args)
        A = torch.tensor(self.graph.A, dtype=torch.float32, requires_grad=False)
        self.register_buffer('A', A)
        spatial_kernel_size = A.size(0)
        temporal_kernel_size = 9
        kernel_size = (temporal_kernel_size, spatial_kernel_size)
        self.num_point = num_point
        self.data_bn = nn.BatchNorm1d(num_person * in_channels * num_point)
        # (Cin, Cout, temporal stride) of the ten blocks, reference :140-151; the first one has no residual
        c = 64
        plan = [(in_channels, c, 1)] + [(c, c, 1)] * 3 + [(c, 2 * c, 2)] + [(2 * c, 2 * c, 1)] * 2 + [(2 * c, 4 * c, 2)] + [(4 * c, 4 * c, 1)] * 2
        self.st_gcn_networks = nn.ModuleList(
            st_gcn(ci, co, kernel_size, s, **(dict(kwargs, residual=False) if i == 0 else kwargs)) for i, (ci, co, s) in enumerate(plan))
        if edge_importance_weighting:
            self.edge_importance = nn.ParameterList([nn.Parameter(torch.ones(self.A.size())) for _ in self.st_gcn_networks])
        else:
            self.edge_importance = [1] * len(self.st_gcn_networks)
        self.fcn = nn.Conv2d(256, num_class, kernel_size=1)
        self.drop_out = nn.Dropout(dropout) if dropout else (lambda x: x)
        Fn.tag_batchnorms_(self)

    def _blocks(self, x):
        if len(x.shape) == 3:
            N, T, VC = x.shape
            x = x.view(N, T, self.num_point, -1).permute(0, 3, 1, 2).contiguous().unsqueeze(-1)
        N, C, T, V, M = x.size()
        x = Fn.StemFn.run(self.data_bn, x, self.data_bn.weight, self.data_bn.bias)      # reference :180-186
        for gcn, importance in zip(self.st_gcn_networks, self.edge_importance):
            x, _ = gcn(x, self.A * importance)
        return x, N, M

    def forward(self, x):
        x, N, M = self._blocks(_require_hip(x))
        if isinstance(self.drop_out, nn.Dropout):
            x = x.view(N, M, x.size(1), -1).mean(3).mean(1)
            x = self.drop_out(x)
            return torch.nn.functional.linear(x, self.fcn.weight.view(self.fcn.weight.size(0), -1), self.fcn.bias)
        return torch.ops.tamgcn.head(x, self.fcn.weight.view(self.fcn.weight.size(0), -1), self.fcn.bias, M)   # :193-198

    def extract_feature(self, x):
        x, N, M = self._blocks(_require_hip(x))
        _, c, t, v = x.size()
        feature = x.view(N, M, c, t, v).permute(0, 2, 3, 4, 1)
        o = torch.ops.tamgcn.pointwise_conv(x, self.fcn.weight, self.fcn.bias)
        output = o.view(N, M, -1, t, v).permute(0, 2, 3, 4, 1)
        return output, feature

    def get_edge_importance_per_joint(self):
        """Mean incoming + outgoing edge weight per joint over all layers, normalised to max 1 (reference :224-252)."""
        if not isinstance(self.edge_importance, nn.ParameterList):
            raise AttributeError('edge_importance_weighting=False: there are no edge weights to summarise')
        imp = torch.stack([p.detach() for p in self.edge_importance]).double().cpu()      # (layers, K, V, V)
        scores = (imp.sum((0, 1, 2)) + imp.sum((0, 1, 3))).numpy()                        # column sums + row sums per joint
        return scores / scores.max()
