"""GPU-box tool: error of the GEMM kernels against an fp64 reference, for the current TAMGCN_SPLIT_BF16 mode
(0 = fp32-input MFMA, 2 = split-fp32 on the bf16 matrix cores).  Run once per mode and compare."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tam_gcn_amd import ops
from tam_gcn_amd.ops import S
dev = torch.device('cuda:0')
torch.manual_seed(0)
print('TAMGCN_SPLIT_BF16 =', os.environ.get('TAMGCN_SPLIT_BF16', '(default 1)'))
def rep(name, got, ref):
    d = (got.double() - ref).abs()
    print(f'  {name:34s} max|err|/max|ref| {float(d.max() / ref.abs().max()):.2e}   rel-L2 {float(d.norm() / ref.norm()):.2e}')
for N, M, K, T in [(64, 192, 64, 64), (64, 384, 128, 32), (64, 768, 256, 16), (64, 256, 256, 16)]:
    V = 20
    gy = torch.randn(N, M, T, V, device=dev); x = torch.randn(N, K, T, V, device=dev).relu_()
    print(f'N={N} M={M} K={K} T={T}')
    dw = ops.wgrad(S(gy), S(x), M=M, K=K).view(M, K)
    rep('wgrad', dw, torch.einsum('nmtv,nktv->mk', gy.double(), x.double()))
    w = torch.randn(M, K, 1, 1, device=dev) * (1.0 / K ** 0.5)
    y, _ = ops.conv(S(x), K=K, w=w, bias=None, M=M)
    rep('conv fwd', y, torch.einsum('mk,nktv->nmtv', w[:, :, 0, 0].double(), x.double()))
    dx, _ = ops.conv(S(gy), K=M, w=w, bias=None, M=K, wmode=1)
    rep('conv bwd-data', dx, torch.einsum('mk,nmtv->nktv', w[:, :, 0, 0].double(), gy.double()))

# fused CTRGC forward: y and the kept x3 against fp64, full and ragged frame chunks
for N, Cin, Cout, T in [(8, 64, 64, 64), (8, 64, 128, 13), (8, 3, 64, 52), (8, 256, 256, 16)]:
    V, S_ = 20, 3
    R = 8 if Cin in (3, 9) else Cin // 8
    x = torch.randn(N, Cin, T, V, device=dev)
    pq = torch.randn(S_ * 2 * R, N, V, device=dev)
    W3 = torch.randn(S_ * Cout, Cin, device=dev) / Cin ** 0.5; B3 = torch.randn(S_ * Cout, device=dev) * 0.1
    W4 = torch.randn(S_, Cout, R, device=dev) / R ** 0.5; B4 = torch.randn(S_, Cout, device=dev) * 0.1
    A = torch.randn(S_, V, V, device=dev) * 0.3; al = torch.tensor([0.7], device=dev)
    y, _, x3 = ops.ctrgc_fwd(S(x), pq, W3, B3, W4, B4, A, al, Cin, Cout, S_, R, stats=False, keep_x3=True)
    x3r = torch.einsum('oc,nctv->notv', W3.double(), x.double()) + B3.double()[None, :, None, None]
    yr = 0
    for s in range(S_):
        p = pq[(2 * s) * R:(2 * s + 1) * R].permute(1, 0, 2).double(); q = pq[(2 * s + 1) * R:(2 * s + 2) * R].permute(1, 0, 2).double()
        E = al.double() * (torch.einsum('cr,nruv->ncuv', W4[s].double(), torch.tanh(p.unsqueeze(-1) - q.unsqueeze(-2))) + B4[s].double()[None, :, None, None]) + A[s].double()
        yr = yr + torch.einsum('ncuv,nctv->nctu', E, x3r[:, s * Cout:(s + 1) * Cout])
    print(f'ctrgc N={N} Cin={Cin} Cout={Cout} T={T}')
    rep('x3 (kept)', x3, x3r); rep('y', y, yr)
    bad = ((x3.double() - x3r).abs() > 1e-3 * x3r.abs().max()).nonzero()
    if len(bad):
        print('   entries of x3 off by > 1e-3 of scale:', len(bad), 'first', bad[:4].tolist())
