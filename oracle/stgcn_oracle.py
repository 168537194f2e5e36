"""ORACLE -- test infrastructure only, never the product path.

CPU restatement (stock PyTorch, functional over a flat state-dict) of the reference's ST-GCN,
/root/reference/models/stgcn.py: ConvTemporalGraphical :37-64, st_gcn :67-99, Model.forward :170-198,
Model.extract_feature :200-222.  Pinned by tests/golden/stgcn.npz, generated from the reference itself by
tests/golden/make_golden_stgcn.py (tests/test_stgcn_oracle.py).  Only tests/ may import this file."""
import torch
import torch.nn.functional as F

from .ctrgcn_oracle import _bn


def conv_temporal_graphical(x, sd, pfx, A):
    """:56-64: 1x1 conv Cin -> K*Cout, then einsum('nkctv,kvw->nctw')."""
    K = A.size(0)
    x = F.conv2d(x, sd[pfx + '.conv.weight'], sd.get(pfx + '.conv.bias'))
    n, kc, t, v = x.size()
    x = x.view(n, K, kc // K, t, v)
    return torch.einsum('nkctv,kvw->nctw', (x, A)).contiguous()


def st_gcn(x, sd, pfx, A, stride, residual, training):
    """:94-99.  residual: 'zero' | 'identity' | 'conv' (:82-91)."""
    if residual == 'zero':
        res = 0
    elif residual == 'identity':
        res = x
    else:
        res = _bn(F.conv2d(x, sd[pfx + '.residual.0.weight'], sd[pfx + '.residual.0.bias'], stride=(stride, 1)), sd, pfx + '.residual.1', training)
    y = conv_temporal_graphical(x, sd, pfx + '.gcn', A)
    y = torch.relu(_bn(y, sd, pfx + '.tcn.0', training))                 # :75-76
    w = sd[pfx + '.tcn.2.weight']
    y = F.conv2d(y, w, sd[pfx + '.tcn.2.bias'], stride=(stride, 1), padding=((w.shape[2] - 1) // 2, 0))
    y = _bn(y, sd, pfx + '.tcn.3', training)                            # dropout 0
    return torch.relu(y + res)


PLAN = [(64, 1, 'zero'), (64, 1, 'identity'), (64, 1, 'identity'), (64, 1, 'identity'), (128, 2, 'conv'), (128, 1, 'identity'),
        (128, 1, 'identity'), (256, 2, 'conv'), (256, 1, 'identity'), (256, 1, 'identity')]           # :138-149


def _blocks(x, sd, num_point, training):
    if x.dim() == 3:
        N, T, VC = x.shape
        x = x.view(N, T, num_point, -1).permute(0, 3, 1, 2).contiguous().unsqueeze(-1)
    N, C, T, V, M = x.size()
    x = x.permute(0, 4, 3, 1, 2).contiguous().view(N * M, V * C, T)
    x = _bn(x, sd, 'data_bn', training)
    x = x.view(N, M, V, C, T).permute(0, 1, 3, 4, 2).contiguous().view(N * M, C, T, V)
    A = sd['A']
    for i, (_, stride, res) in enumerate(PLAN):
        imp = sd.get(f'edge_importance.{i}')
        x = st_gcn(x, sd, f'st_gcn_networks.{i}', A * imp if imp is not None else A, stride, res, training)
    return x, N, M


def model_forward(x, sd, num_point, training=True):
    x, N, M = _blocks(x, sd, num_point, training)
    x = F.avg_pool2d(x, x.size()[2:])
    x = x.view(N, M, -1, 1, 1).mean(dim=1)
    x = F.conv2d(x, sd['fcn.weight'], sd['fcn.bias'])
    return x.view(x.size(0), -1)


def model_extract_feature(x, sd, num_point, training=False):
    x, N, M = _blocks(x, sd, num_point, training)
    _, c, t, v = x.size()
    feature = x.view(N, M, c, t, v).permute(0, 2, 3, 4, 1)
    o = F.conv2d(x, sd['fcn.weight'], sd['fcn.bias'])
    return o.view(N, M, -1, t, v).permute(0, 2, 3, 4, 1), feature
