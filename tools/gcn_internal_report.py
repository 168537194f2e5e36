"""GPU-box diagnostic: step-by-step comparison of gcn_forward/gcn_backward intermediates for one
block of a model case against an fp64 replica fed with the same (fp64-exact) inputs."""
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
from tam_gcn_amd import functional as Fn, ops                                       # noqa: E402
from tam_gcn_amd.ops import S                                                       # noqa: E402
from oracle import ctrgcn_oracle as O                                               # noqa: E402

dev = torch.device('cuda:0')
tagsel, layer = sys.argv[1], int(sys.argv[2])


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def bn64(x, w, b):
    mean = x.mean((0, 2, 3), keepdim=True)
    var = x.var((0, 2, 3), unbiased=False, keepdim=True)
    return (x - mean) / torch.sqrt(var + 1e-5) * w[None, :, None, None] + b[None, :, None, None]


for tag, margs, shape in MODEL_CASES:
    if tag != tagsel:
        continue
    m = M.Model(**margs)
    fill_state_(m.state_dict(), seed=MODEL_PARAM_SEED)
    sd = {k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in m.state_dict().items()}
    x = make_input(shape, seed=MODEL_X_SEED).double()
    lab = make_labels(shape[0], margs['num_class'], seed=MODEL_LABEL_SEED)
    # run the fp64 oracle to get the block input and the gradient wrt the gcn output
    for k, v in sd.items():
        if v.is_floating_point() and 'running' not in k:
            v.requires_grad_(True)
    h, N, Mp = O._stem(x, sd, margs['num_point'], True)
    keep = {}
    for i in range(1, 11):
        pfx = f'l{i}'
        stride = O._STRIDES.get(i, 1)
        xin = h
        g = O.unit_gcn(xin, sd, pfx + '.gcn1', True)
        if i == layer:
            g.retain_grad(); keep['xin'], keep['g'] = xin, g
        yt = O.ms_tcn(g, sd, pfx + '.tcn1', 5, stride, (1, 2), True, 'zero')
        r = 0 if i == 1 else (O.unit_tcn(xin, sd, pfx + '.residual', 1, stride, True)
                              if (pfx + '.residual.conv.weight') in sd else xin)
        h = torch.relu(yt + r)
    feat = h.view(N, Mp, h.size(1), -1).mean(3).mean(1)
    F.cross_entropy(F.linear(feat, sd['fc.weight'], sd['fc.bias']), lab).backward()
    xin, dgr = keep['xin'].detach(), keep['g'].grad.detach()

    # ---- fp64 replica of unit_gcn with every intermediate exposed
    p = f'l{layer}.gcn1'
    P64 = {k[len(p) + 1:]: v.detach() for k, v in sd.items() if k.startswith(p + '.')}
    xr = xin.clone().requires_grad_(True)
    PA, alpha = P64['PA'], P64['alpha']
    ypre = 0
    x3s = []
    for s in range(3):
        c = f'convs.{s}.'
        w = lambda n: P64[c + n + '.weight'][:, :, 0, 0]
        bb = lambda n: P64[c + n + '.bias']
        pq_p = torch.einsum('rc,nctv->nrtv', w('conv1'), xr).mean(2) + bb('conv1')[None, :, None]
        pq_q = torch.einsum('rc,nctv->nrtv', w('conv2'), xr).mean(2) + bb('conv2')[None, :, None]
        D = torch.tanh(pq_p.unsqueeze(-1) - pq_q.unsqueeze(-2))
        E = alpha * (torch.einsum('cr,nruv->ncuv', w('conv4'), D) + bb('conv4')[None, :, None, None]) + PA[s][None, None]
        x3 = torch.einsum('oc,nctv->notv', w('conv3'), xr) + bb('conv3')[None, :, None, None]
        x3.retain_grad(); x3s.append(x3)
        ypre = ypre + torch.einsum('ncuv,nctv->nctu', E, x3)
    ypre.retain_grad()
    ybn = bn64(ypre, P64['bn.weight'], P64['bn.bias']); ybn.retain_grad()
    if 'down.0.weight' in P64:
        dpre = torch.einsum('oc,nctv->notv', P64['down.0.weight'][:, :, 0, 0], xr) + P64['down.0.bias'][None, :, None, None]
        dpre.retain_grad()
        res = bn64(dpre, P64['down.1.weight'], P64['down.1.bias'])
    else:
        dpre, res = None, (xr if layer != 1 else 0)
    diff = res - ybn
    opre = torch.einsum('oc,nctv->notv', P64['offset_conv.0.weight'][:, :, 0, 0], diff) + P64['offset_conv.0.bias'][None, :, None, None]
    opre.retain_grad()
    obn = bn64(opre, P64['offset_conv.1.weight'], P64['offset_conv.1.bias']); obn.retain_grad()
    off = torch.tanh(obn)
    gsum = ybn + off + res
    gsum.retain_grad()
    g64 = torch.relu(gsum)
    g64.backward(dgr)

    # ---- HIP path, same inputs
    blk = getattr(m, f'l{layer}').gcn1.to(dev).train()
    with torch.no_grad():
        xg = xin.float().to(dev).contiguous()
        Pk = blk._pack([t.detach() for t in blk._tensors(dev)])
        gg, sv = Fn.gcn_forward(xg, Pk, True, save=True)
        print(f'{tag} l{layer}: fwd  y_pre {rel(sv["y_pre"], ypre):.1e}  o_pre {rel(sv["o_pre"], opre):.1e}  g {rel(gg, g64):.1e}')
        ybn_h = ops.apply(S(sv['y_pre'], coef=sv['coef_y']), Pk.Cout)
        obn_h = ops.apply(S(sv['o_pre'], coef=sv['coef_o']), Pk.Cout)
        print(f'   ybn {rel(ybn_h, ybn):.1e}  obn {rel(obn_h, obn):.1e}  |obn|max {float(obn.abs().max()):.2f}')
        dg = dgr.float().to(dev).contiguous()
        dsum, doz, part_o = ops.gcn_tail_bwd(dg, gg, S(sv['o_pre'], coef=sv['coef_o']), sv['save_o'])
        print(f'   dsum {rel(dsum, gsum.grad):.1e}  doz(d obn) {rel(doz, obn.grad):.1e}')
        coefb_o = torch.empty(3, Pk.Cout, device=dev)
        Pk.bno.bwd(part_o, 0, xg.shape[0] * xg.shape[2] * xg.shape[3], sv['save_o'], 0, True, coefb_o, 0, True)
        dopre_h = ops.apply(S(doz, sv['o_pre'], coefb_o), Pk.Cout)
        print(f'   d o_pre {rel(dopre_h, opre.grad):.1e}')
        ddiff, _ = ops.conv(S(doz, sv['o_pre'], coefb_o), K=Pk.Cout, w=Pk.Wo, bias=None, M=Pk.Cout, wmode=1)
        ddiff64 = torch.einsum('oc,notv->nctv', P64['offset_conv.0.weight'][:, :, 0, 0], opre.grad)
        print(f'   ddiff {rel(ddiff, ddiff64):.1e}')
        dyb, dres, part2 = ops.gcn_mid_bwd(dsum, ddiff, sv['y_pre'], sv['save_y'], sv['d_pre'], sv['save_d'], want_dres=Pk.mode != 'zero')
        print(f'   dyb (d ybn) {rel(dyb, ybn.grad):.1e}')
        coefb_y = torch.empty(3, Pk.Cout, device=dev)
        Pk.bn.bwd(part2, 0, xg.shape[0] * xg.shape[2] * xg.shape[3], sv['save_y'], 0, True, coefb_y, 0)
        dypre_h = ops.apply(S(dyb, sv['y_pre'], coefb_y), Pk.Cout)
        print(f'   d y_pre {rel(dypre_h, ypre.grad):.1e}   (coef_y c1 range {float(sv["coef_y"][0].abs().min()):.2e}..{float(sv["coef_y"][0].abs().max()):.2e})')
        dx, G = Fn.gcn_backward(Pk, sv, dg, need_dx=True)
        dx3_64 = torch.cat([t.grad for t in x3s], 1)
        print(f'   dx {rel(dx, xr.grad):.1e}  dPA {rel(G["PA"], torch.zeros(1)) if False else 0:.0f}')
        # per-channel view of where d y_pre is off
        e = (dypre_h.cpu().double() - ypre.grad).abs().amax((0, 2, 3)) / ypre.grad.abs().max()
        worst = torch.topk(e, 5)
        print('   worst channels of d y_pre:', [(int(i), f'{float(v):.1e}') for v, i in zip(worst.values, worst.indices)])
        invstd = sv['save_y'][1].cpu()
        print('   invstd of those:', [f'{float(invstd[int(i)]):.3e}' for i in worst.indices],
              ' invstd range', f'{float(invstd.min()):.2e}..{float(invstd.max()):.2e}')
        e2 = (dopre_h.cpu().double() - opre.grad).abs().amax((0, 2, 3)) / opre.grad.abs().max()
        w2 = torch.topk(e2, 5)
        print('   worst channels of d o_pre:', [(int(i), f'{float(v):.1e}') for v, i in zip(w2.values, w2.indices)],
              ' invstd_o', [f'{float(sv["save_o"][1][int(i)]):.2e}' for i in w2.indices])
