"""GPU-box diagnostic: unit-mode (TCN_GCN_unit) backward internals for one block of a model case:
dz, dg, dxres from tcn_backward vs fp64, with the location and count of the bad elements."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from cases import MODEL_CASES, MODEL_PARAM_SEED, MODEL_X_SEED, MODEL_LABEL_SEED   # noqa: E402
from params import fill_state_, make_input, make_labels                           # noqa: E402
from tam_gcn_amd.models import ctrgcn as M                                          # noqa: E402
from tam_gcn_amd import functional as Fn                                            # noqa: E402
from oracle import ctrgcn_oracle as O                                               # noqa: E402

dev = torch.device('cuda:0')
tagsel, layer = sys.argv[1], int(sys.argv[2])


def report(name, a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs()
    sc = b.abs().max()
    bad = (err > 1e-4 * sc)
    idx = torch.nonzero(bad)
    print(f'   {name}: max rel err {float(err.max() / sc):.2e}; elements off by >1e-4*max: {int(bad.sum())} of {a.numel()}')
    for row in idx[:6]:
        t = tuple(int(v) for v in row)
        print(f'      at {t}: got {float(a[t]):+.6e} want {float(b[t]):+.6e}')
    return idx


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
    keep = {}
    for i in range(1, 11):
        pfx = f'l{i}'
        stride = O._STRIDES.get(i, 1)
        xin = h
        if i == layer:
            xin.retain_grad()
        g = O.unit_gcn(xin, sd, pfx + '.gcn1', True)
        yt = O.ms_tcn(g, sd, pfx + '.tcn1', 5, stride, (1, 2), True, 'zero')
        r = 0 if i == 1 else (O.unit_tcn(xin, sd, pfx + '.residual', 1, stride, True)
                              if (pfx + '.residual.conv.weight') in sd else xin)
        pre = yt + r
        h = torch.relu(pre)
        if i == layer:
            g.retain_grad(); yt.retain_grad(); pre.retain_grad(); h.retain_grad()
            keep = dict(xin=xin, g=g, yt=yt, pre=pre, h=h)
    feat = h.view(N, Mp, h.size(1), -1).mean(3).mean(1)
    F.cross_entropy(F.linear(feat, sd['fc.weight'], sd['fc.bias']), lab).backward()
    blk = getattr(m, f'l{layer}').to(dev).train()
    with torch.no_grad():
        f32 = lambda t: t.detach().float().to(dev).contiguous()
        xg, gg = f32(keep['xin']), f32(keep['g'])
        Pt = blk._pack_tcn([t.detach() for t in (blk.tcn1._tensors() + ([blk.residual.conv.weight, blk.residual.conv.bias, blk.residual.bn.weight, blk.residual.bn.bias] if blk._rmode == 'conv' else []))])
        out, svt = Fn.tcn_forward(gg, Pt, True, True, xres=xg)
        print(f'{tag} l{layer} (rmode {Pt.rmode}, relu {Pt.relu}):')
        report('out', out, keep['h'])
        pre64 = keep['pre'].detach()
        near = (pre64.abs() < 1e-5 * pre64.abs().max()).sum()
        print(f'   pre-activation |.| < 1e-5*max: {int(near)} elements; exact zeros in x: {int((keep["xin"] == 0).sum())}')
        dout = f32(keep['h'].grad)
        dg, dxres, Gt = Fn.tcn_backward(Pt, svt, dout, need_dg=True, need_dxres=True)
        report('dz (=d pre)', dxres if Pt.rmode == 'identity' else dxres, keep['pre'].grad if Pt.rmode == 'identity' else keep['xin'].grad * 0 + 0) if Pt.rmode == 'identity' else None
        report('dg', dg, keep['g'].grad)
        # ---- the full chain exactly as TCNGCNUnitFn runs it (tcn consumes the HIP gcn output)
        Pg = blk.gcn1._pack([t.detach() for t in blk.gcn1._tensors(dev)])
        g_h, svg = Fn.gcn_forward(xg, Pg, True, True)
        out2, svt2 = Fn.tcn_forward(g_h, Pt, True, True, xres=xg)
        m_g = (g_h > 0).cpu() != (keep['g'].detach() > 0)
        m_o = (out2 > 0).cpu() != (keep['h'].detach() > 0)
        print(f'   chain: relu-mask flips vs fp64: gcn output {int(m_g.sum())}, block output {int(m_o.sum())}')
        for mm, nm, ref in ((m_g, 'g', keep['g']), (m_o, 'out', keep['h'])):
            for row in torch.nonzero(mm)[:4]:
                t = tuple(int(v) for v in row)
                print(f'      flip in {nm} at {t}: fp64 value {float(ref[t]):+.3e}, upstream grad there {float(ref.grad[t]):+.3e} (max |grad| {float(ref.grad.abs().max()):.3e})')
        dg2, dxres2, _ = Fn.tcn_backward(Pt, svt2, dout, need_dg=True, need_dxres=True)
        report('chain dg', dg2, keep['g'].grad)
        dx2, G2 = Fn.gcn_backward(Pg, svg, dg2, need_dx=True, extra_dx=dxres2 if Pt.rmode != 'zero' else None)
        report('chain dx', dx2, keep['xin'].grad)
        report('chain dPA', G2['PA'], sd[f'l{layer}.gcn1.PA'].grad)
        report('chain dWo', G2['Wo'], sd[f'l{layer}.gcn1.offset_conv.0.weight'].grad)
