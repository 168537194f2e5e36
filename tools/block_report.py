"""GPU-box diagnostic: isolates every TCN_GCN_unit (and its gcn1 / tcn1 halves) of a model case.
The fp64 CPU oracle provides each block's input, intermediate g, upstream gradients and the
exact results; the HIP modules are fed the fp32 casts.  Prints relative errors (fp32 noise ~1e-6)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from cases import MODEL_CASES, MODEL_PARAM_SEED, MODEL_X_SEED, MODEL_LABEL_SEED   # noqa: E402
from params import fill_state_, make_input, make_labels                           # noqa: E402
from tam_gcn_amd.models import ctrgcn as M                                          # noqa: E402
from oracle import ctrgcn_oracle as O                                               # noqa: E402

dev = torch.device('cuda:0')


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


tagsel = sys.argv[1] if len(sys.argv) > 1 else 'ucla_t64'
for tag, margs, shape in MODEL_CASES:
    if tag != tagsel:
        continue
    m = M.Model(**margs)
    fill_state_(m.state_dict(), seed=MODEL_PARAM_SEED)
    sd = {k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in m.state_dict().items()}
    for k, v in sd.items():
        if v.is_floating_point() and 'running' not in k:
            v.requires_grad_(True)
    x = make_input(shape, seed=MODEL_X_SEED).double()
    lab = make_labels(shape[0], margs['num_class'], seed=MODEL_LABEL_SEED)
    h, N, Mp = O._stem(x, sd, margs['num_point'], True)
    rec = []
    for i in range(1, 11):
        pfx = f'l{i}'
        stride = O._STRIDES.get(i, 1)
        xin = h
        xin.retain_grad()
        g = O.unit_gcn(xin, sd, pfx + '.gcn1', True)
        g.retain_grad()
        yt = O.ms_tcn(g, sd, pfx + '.tcn1', 5, stride, (1, 2), True, 'zero')
        yt.retain_grad()
        if i == 1:
            r = 0
        elif (pfx + '.residual.conv.weight') in sd:
            r = O.unit_tcn(xin, sd, pfx + '.residual', 1, stride, True)
        else:
            r = xin
        h = torch.relu(yt + r)
        h.retain_grad()
        rec.append((i, xin, g, yt, h))
    c_new = h.size(1)
    feat = h.view(N, Mp, c_new, -1).mean(3).mean(1)
    logits = torch.nn.functional.linear(feat, sd['fc.weight'], sd['fc.bias'])
    torch.nn.functional.cross_entropy(logits, lab).backward()

    m = m.to(dev).train()
    f32 = lambda t: t.detach().float().to(dev).contiguous()
    for i, xin, g, yt, hout in rec:
        blk = getattr(m, f'l{i}')
        pg = lambda name: sd[f'l{i}.{name}'].grad
        # --- full unit
        xi = f32(xin).requires_grad_(True)
        for p in blk.parameters():
            p.grad = None
        out = blk(xi)
        out.backward(f32(hout.grad))
        e_out, e_dx = rel(out, hout), rel(xi.grad, xin.grad)
        e_pa = rel(blk.gcn1.PA.grad, pg('gcn1.PA'))
        e_w3 = rel(blk.gcn1.convs[0].conv3.weight.grad, pg('gcn1.convs.0.conv3.weight'))
        e_wo = rel(blk.gcn1.offset_conv[0].weight.grad, pg('gcn1.offset_conv.0.weight'))
        e_wt = rel(blk.tcn1.branches[1][3].conv.weight.grad, pg('tcn1.branches.1.3.conv.weight'))
        e_win = rel(blk.tcn1.branches[0][0].weight.grad, pg('tcn1.branches.0.0.weight'))
        # --- gcn1 alone
        xi2 = f32(xin).requires_grad_(True)
        g2 = blk.gcn1(xi2)
        g2.backward(f32(g.grad))
        # gcn part of dx: total dx minus residual path is not separable in the oracle; report g and PA
        e_g = rel(g2, g)
        # --- tcn1 alone
        gi = f32(g).requires_grad_(True)
        y2 = blk.tcn1(gi)
        y2.backward(f32(yt.grad))
        e_yt, e_dg = rel(y2, yt), rel(gi.grad, g.grad)
        print(f'l{i:<2d} unit: out {e_out:.1e} dx {e_dx:.1e} dPA {e_pa:.1e} dW3 {e_w3:.1e} dWo {e_wo:.1e} dWt {e_wt:.1e} dWin {e_win:.1e}'
              f' | gcn1: g {e_g:.1e} | tcn1: y {e_yt:.1e} dg {e_dg:.1e}')
