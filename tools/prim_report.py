"""GPU-box diagnostic: fused CTRGC fwd/bwd primitives against an fp64 CPU evaluation; prints the
relative error of every output (expected ~1e-6 for fp32 kernels)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from params import make_input                       # noqa: E402
from tam_gcn_amd import ops                         # noqa: E402
from tam_gcn_amd.ops import S                       # noqa: E402

d = torch.device('cuda:0')


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def run(N, Cin, Cout, T, V, S_):
    R = 8 if Cin in (3, 9) else Cin // 8
    mk = lambda shape, seed, sc=1.0: (make_input(shape, seed).double() * sc).requires_grad_(True)
    x = mk((N, Cin, T, V), 1)
    W12 = mk((S_ * 2 * R, Cin), 2, 1.0 / Cin ** 0.5); B12 = mk((S_ * 2 * R,), 3, 0.1)
    W3 = mk((S_ * Cout, Cin), 4, 1.0 / Cin ** 0.5); B3 = mk((S_ * Cout,), 5, 0.1)
    W4 = mk((S_, Cout, R), 6, 1.0 / R ** 0.5); B4 = mk((S_, Cout), 7, 0.1)
    A = mk((S_, V, V), 8, 0.3)
    alpha = torch.tensor([0.7], dtype=torch.float64, requires_grad=True)
    xbar = x.mean(2)
    pqr = torch.einsum('jc,ncv->jnv', W12, xbar) + B12[:, None, None]
    pqr.retain_grad()
    y = 0
    for s in range(S_):
        p = pqr[(2 * s) * R:(2 * s + 1) * R].permute(1, 0, 2)
        q = pqr[(2 * s + 1) * R:(2 * s + 2) * R].permute(1, 0, 2)
        D = torch.tanh(p.unsqueeze(-1) - q.unsqueeze(-2))
        E = alpha * (torch.einsum('cr,nruv->ncuv', W4[s], D) + B4[s][None, :, None, None]) + A[s][None, None]
        x3 = torch.einsum('oc,nctv->notv', W3[s * Cout:(s + 1) * Cout], x) + B3[s * Cout:(s + 1) * Cout][None, :, None, None]
        y = y + torch.einsum('ncuv,nctv->nctu', E, x3)
    cot = make_input(tuple(y.shape), 9).double()
    (y * cot).sum().backward()
    t = lambda z: z.detach().float().to(d).contiguous()
    xs = S(t(x))
    xb = ops.tmean(xs, Cin)
    pq, _ = ops.conv(S(xb.view(1, Cin, N, V)), K=Cin, w=t(W12), bias=t(B12), M=S_ * 2 * R)
    pq = pq.view(S_ * 2 * R, N, V)
    yg, part, x3k = ops.ctrgc_fwd(xs, pq, t(W3), t(B3), t(W4), t(B4), t(A), t(alpha), Cin, Cout, S_, R, stats=True)
    dx3, db3, dA, dW4, db4, dal, dpq = ops.ctrgc_bwd(xs, pq, t(W3), t(B3), t(W4), t(B4), t(A), t(alpha),
                                                     Cin, Cout, S_, R, S(t(cot)))
    dW3 = ops.wgrad(S(dx3), xs, M=S_ * Cout, K=Cin)
    dpq4 = S(dpq.view(1, S_ * 2 * R, N, V))
    dxbar, _ = ops.conv(dpq4, K=S_ * 2 * R, w=t(W12), bias=None, M=Cin, wmode=1)
    dx, _ = ops.conv(S(dx3), K=S_ * Cout, w=t(W3), bias=None, M=Cin, wmode=1, bcast=dxbar.view(Cin, N, V),
                     bcast_scale=1.0 / T)
    dW12 = ops.wgrad(dpq4, S(xb.view(1, Cin, N, V)), M=S_ * 2 * R, K=Cin)
    torch.cuda.synchronize()
    # reference pieces
    dxbar_ref = torch.einsum('jc,jnv->cnv', W12.detach(), pqr.grad)
    print(f'-- N={N} Cin={Cin} Cout={Cout} T={T} V={V} S={S_} R={R}')
    for nm, a, b in [('xbar', xb, xbar.permute(1, 0, 2)), ('pq', pq, pqr), ('y', yg, y), ('dA', dA, A.grad), ('dW4', dW4, W4.grad),
                     ('db4', db4, B4.grad), ('dalpha', dal, alpha.grad), ('db3', db3, B3.grad), ('dpq', dpq, pqr.grad),
                     ('dW3', dW3.view(S_ * Cout, Cin), W3.grad), ('dxbar', dxbar.view(Cin, N, V), dxbar_ref),
                     ('dx', dx, x.grad), ('dW12', dW12.view(S_ * 2 * R, Cin), W12.grad), ('dB12', dpq.sum((1, 2)), B12.grad)]:
        print(f'   {nm:7s} rel err {rel(a, b):.2e}')


for shp in [(2, 64, 64, 20, 20, 3), (2, 256, 256, 16, 20, 3), (4, 256, 256, 16, 20, 3), (1, 128, 128, 5, 25, 3), (4, 64, 128, 20, 25, 3),
            (3, 3, 64, 9, 20, 3)]:
    run(*shp)
