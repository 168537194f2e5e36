"""ORACLE — test infrastructure only, never the product path.

CPU restatement (stock PyTorch fp32 ops, functional style over a flat
state-dict) of the reference's CTR-GCN hot path, /root/reference/models/ctrgcn.py.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this file; ``tam_gcn_amd`` never does.

Parity pinning: every function below is checked against golden vectors that
``tests/golden/make_golden.py`` produced by importing the reference itself in
the build container (tests/test_oracle_vs_golden.py).  The reference publishes
no golden vectors of its own.

All functions take ``sd`` (a mapping name -> tensor using the reference's
state-dict key names) and a key prefix.  Train-mode BatchNorm
uses batch statistics and writes the updated running statistics into ``sd``
in place, exactly as ``nn.BatchNorm2d`` does (momentum 0.1, eps 1e-5,
unbiased running variance).
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------
def _conv(x, sd, pfx, stride=1, pad=0, dil=1):
    """k x 1 convolution over (N, C, T, V); weight (O, I, k, 1)."""
    return F.conv2d(x, sd[pfx + '.weight'], sd.get(pfx + '.bias'),
                    stride=(stride, 1), padding=(pad, 0), dilation=(dil, 1))


def _bn(x, sd, pfx, training):
    """BatchNorm over channel dim 1 (works for 3-D and 4-D input)."""
    rm, rv = sd[pfx + '.running_mean'], sd[pfx + '.running_var']
    out = F.batch_norm(x, rm, rv, sd[pfx + '.weight'], sd[pfx + '.bias'],
                       training=training, momentum=BN_MOMENTUM, eps=BN_EPS)
    if training and (pfx + '.num_batches_tracked') in sd:
        sd[pfx + '.num_batches_tracked'] += 1
    return out


# --------------------------------------------------------------------------
# CTRGC  (reference: models/ctrgcn.py:150-177)
# --------------------------------------------------------------------------
def ctrgc(x, sd, pfx, A=None, alpha=1):
    """x (N,Cin,T,V) -> (N,Cout,T,V).  models/ctrgcn.py:172-177.

    p/q: 1x1 conv over the full clip, then mean over T (:173);
    D = tanh(p_u - q_v) (:174); E = conv4(D)*alpha + A (:175);
    out[n,c,t,u] = sum_v E[n,c,u,v] * conv3(x)[n,c,t,v] (:176).
    """
    p = _conv(x, sd, pfx + '.conv1').mean(-2)
    q = _conv(x, sd, pfx + '.conv2').mean(-2)
    x3 = _conv(x, sd, pfx + '.conv3')
    D = torch.tanh(p.unsqueeze(-1) - q.unsqueeze(-2))            # N,R,V,V
    E = _conv(D, sd, pfx + '.conv4') * alpha
    if A is not None:
        E = E + A[None, None]
    return torch.einsum('ncuv,nctv->nctu', E, x3)


def ctrgc_mean_commuted(x, sd, pfx, A, alpha):
    """Algebraic restatement used as the kernel specification: the 1x1 convs
    commute with the mean over T.  Differs from ``ctrgc``
    only by fp32 rounding (~1e-6 relative)."""
    xbar = x.mean(2, keepdim=True)                               # N,Cin,1,V
    p = _conv(xbar, sd, pfx + '.conv1')[:, :, 0]
    q = _conv(xbar, sd, pfx + '.conv2')[:, :, 0]
    x3 = _conv(x, sd, pfx + '.conv3')
    D = torch.tanh(p.unsqueeze(-1) - q.unsqueeze(-2))
    E = torch.einsum('cr,nruv->ncuv', sd[pfx + '.conv4.weight'][:, :, 0, 0], D)
    E = (E + sd[pfx + '.conv4.bias'][None, :, None, None]) * alpha + A[None, None]
    return torch.einsum('ncuv,nctv->nctu', E, x3)


# --------------------------------------------------------------------------
# unit_gcn  (reference: models/ctrgcn.py:196-263; this is the *modified*
# block with the offset_conv branch, :219-223, :256-259)
# --------------------------------------------------------------------------
def unit_gcn(x, sd, pfx, training=True):
    PA, alpha = sd[pfx + '.PA'], sd[pfx + '.alpha']
    y = None
    for i in range(PA.shape[0]):                                  # :252-254
        z = ctrgc(x, sd, f'{pfx}.convs.{i}', PA[i], alpha)
        y = z if y is None else z + y
    y = _bn(y, sd, pfx + '.bn', training)                         # :255
    if (pfx + '.down.0.weight') in sd:                            # :209-214
        res = _bn(_conv(x, sd, pfx + '.down.0'), sd, pfx + '.down.1', training)
    else:
        res = x                                                   # :216
    diff = res - y                                                # :257
    off = torch.tanh(_bn(_conv(diff, sd, pfx + '.offset_conv.0'), sd,
                         pfx + '.offset_conv.1', training))       # :219-223,258
    return torch.relu(y + off + res)                              # :259-261


def unit_gcn_noresidual(x, sd, pfx, training=True):
    """unit_gcn(residual=False): ``down`` is the constant 0 (:217-218)."""
    PA, alpha = sd[pfx + '.PA'], sd[pfx + '.alpha']
    y = sum(ctrgc(x, sd, f'{pfx}.convs.{i}', PA[i], alpha) for i in range(PA.shape[0]))
    y = _bn(y, sd, pfx + '.bn', training)
    off = torch.tanh(_bn(_conv(0 - y, sd, pfx + '.offset_conv.0'), sd,
                         pfx + '.offset_conv.1', training))
    return torch.relu(y + off)


# --------------------------------------------------------------------------
# temporal side (reference: models/ctrgcn.py:52-69, :72-147, :179-193)
# --------------------------------------------------------------------------
def temporal_conv(x, sd, pfx, kernel_size, stride=1, dilation=1, training=True):
    """TemporalConv: dilated k x 1 conv + BN.  pad = (k+(k-1)(d-1)-1)//2 (:55)."""
    pad = (kernel_size + (kernel_size - 1) * (dilation - 1) - 1) // 2
    return _bn(_conv(x, sd, pfx + '.conv', stride, pad, dilation), sd, pfx + '.bn', training)


def unit_tcn(x, sd, pfx, kernel_size=9, stride=1, training=True):
    """unit_tcn: BN(conv k x 1), no ReLU applied (:191-193)."""
    pad = int((kernel_size - 1) / 2)
    return _bn(_conv(x, sd, pfx + '.conv', stride, pad, 1), sd, pfx + '.bn', training)


def ms_tcn(x, sd, pfx, kernel_size=3, stride=1, dilations=(1, 2, 3, 4),
           training=True, residual='zero', residual_kernel_size=1):
    """MultiScale_TemporalConv (:72-147).  ``residual``: 'zero' (residual=False,
    the only form TCN_GCN_unit uses, :270-271), 'identity', or 'conv'."""
    nb = len(dilations)
    ks = kernel_size if isinstance(kernel_size, (list, tuple)) else [kernel_size] * nb
    outs = []
    for b, (k, d) in enumerate(zip(ks, dilations)):               # :93-110
        h = torch.relu(_bn(_conv(x, sd, f'{pfx}.branches.{b}.0'), sd,
                           f'{pfx}.branches.{b}.1', training))
        outs.append(temporal_conv(h, sd, f'{pfx}.branches.{b}.3', k, stride, d, training))
    b = nb                                                        # :113-119
    h = torch.relu(_bn(_conv(x, sd, f'{pfx}.branches.{b}.0'), sd,
                       f'{pfx}.branches.{b}.1', training))
    h = F.max_pool2d(h, kernel_size=(3, 1), stride=(stride, 1), padding=(1, 0))
    outs.append(_bn(h, sd, f'{pfx}.branches.{b}.4', training))
    b = nb + 1                                                    # :121-124
    outs.append(_bn(_conv(x, sd, f'{pfx}.branches.{b}.0', stride), sd,
                    f'{pfx}.branches.{b}.1', training))
    out = torch.cat(outs, dim=1)                                  # :145
    if residual == 'identity':
        out = out + x
    elif residual == 'conv':
        out = out + temporal_conv(x, sd, pfx + '.residual', residual_kernel_size,
                                  stride, 1, training)
    return out


def tcn_gcn_unit(x, sd, pfx, stride=1, residual=True, training=True,
                 kernel_size=5, dilations=(1, 2)):
    """TCN_GCN_unit (:266-284): relu(tcn1(gcn1(x)) + residual(x))."""
    g = unit_gcn(x, sd, pfx + '.gcn1', training)
    y = ms_tcn(g, sd, pfx + '.tcn1', kernel_size, stride, dilations, training, 'zero')
    if not residual:                                              # :273-274
        r = 0
    elif (pfx + '.residual.conv.weight') in sd:                   # :279-280
        r = unit_tcn(x, sd, pfx + '.residual', 1, stride, training)
    else:
        r = x                                                     # :276-277
    return torch.relu(y + r)


# --------------------------------------------------------------------------
# Model  (reference: models/ctrgcn.py:287-375)
# --------------------------------------------------------------------------
_STRIDES = {5: 2, 8: 2}          # :309, :312


def _stem(x, sd, num_point, training):
    if x.dim() == 3:                                              # :325-327
        N, T, VC = x.shape
        x = x.view(N, T, num_point, -1).permute(0, 3, 1, 2).contiguous().unsqueeze(-1)
    N, C, T, V, M = x.shape
    x = x.permute(0, 4, 3, 1, 2).contiguous().view(N, M * V * C, T)   # :330
    x = _bn(x, sd, 'data_bn', training)                           # :331
    x = x.view(N, M, V, C, T).permute(0, 1, 3, 4, 2).contiguous().view(N * M, C, T, V)
    return x, N, M


def model_blocks(x, sd, num_point, training=True):
    x, N, M = _stem(x, sd, num_point, training)
    for i in range(1, 11):                                        # :333-342
        x = tcn_gcn_unit(x, sd, f'l{i}', _STRIDES.get(i, 1), residual=(i != 1),
                         training=training)
    return x, N, M


def model_forward(x, sd, num_point, training=True):
    """Model.forward (:324-348), drop_out=0."""
    x, N, M = model_blocks(x, sd, num_point, training)
    c_new = x.size(1)
    x = x.view(N, M, c_new, -1).mean(3).mean(1)                   # :343-345
    return F.linear(x, sd['fc.weight'], sd['fc.bias'])            # :348


def model_extract_feature(x, sd, num_point, training=True):
    """Model.extract_feature (:350-375): (N, C, T/4, V, M), returned twice."""
    x, N, M = model_blocks(x, sd, num_point, training)
    NM, C, T, V = x.shape
    x = x.view(N, M, C, T, V).permute(0, 2, 3, 4, 1).contiguous()
    return x, x


# --------------------------------------------------------------------------
# helpers shared by tests / bench
# --------------------------------------------------------------------------
def clone_state(sd, requires_grad=False):
    """Detached fp32 copy on CPU; float parameters optionally require grad."""
    out = {}
    for k, v in sd.items():
        t = v.detach().to('cpu').clone()
        if requires_grad and t.is_floating_point() and 'running_' not in k:
            t.requires_grad_(True)
        out[k] = t
    return out


def degenerate_fix_(sd, seed=0):
    """De-degenerate the reference's default init in place so that every
    branch contributes (SURVEY.md §7 'Degenerate default init', §8d):
    alpha ~ U(0.3,1), unit_gcn.bn.weight ~ N(1,0.02), offset_conv conv
    weight kaiming fan_out, PA += 0.05*N(0,1), all biases ~ 0.05*N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, v in sd.items():
            parts = k.split('.')
            if parts[-1] == 'alpha':
                v.copy_(torch.rand(v.shape, generator=g) * 0.7 + 0.3)
            elif parts[-2:] == ['bn', 'weight'] and (len(parts) == 2 or parts[-3] == 'gcn1'):
                v.copy_(1 + 0.02 * torch.randn(v.shape, generator=g))      # unit_gcn.bn
            elif parts[-3:] == ['offset_conv', '0', 'weight']:
                v.copy_(torch.randn(v.shape, generator=g) * (2.0 / v.shape[0]) ** 0.5)
            elif parts[-1] == 'PA':
                v.add_(0.05 * torch.randn(v.shape, generator=g))
            elif parts[-1] == 'bias' and v.dim() == 1:
                v.add_(0.05 * torch.randn(v.shape, generator=g))
    return sd
