"""GPU-box check: the three-term-split forward GEMM (conv1x1_glds_split_kernel<3, fwd>) against the exact fp32-input MFMA
kernels on the same operands, over the shapes the model produces (small T, partial tiles, one and two sources, moments)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tam_gcn_amd import _lib, ops
from tam_gcn_amd.ops import S
lib = _lib.load()
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
r = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).to(dev)
worst = 0.0
for (N, K, M, T, V, two) in [(4, 64, 128, 7, 20, 0), (4, 128, 128, 7, 20, 1), (4, 128, 256, 4, 20, 0), (4, 256, 256, 4, 20, 1), (2, 256, 768, 16, 20, 0),
                             (3, 64, 192, 13, 20, 0), (2, 128, 384, 33, 25, 0), (2, 256, 768, 12, 64, 0), (256, 256, 256, 16, 20, 1)]:
    x1, x2 = r(N, K, T, V), r(N, K, T, V)
    coef = torch.stack((r(K) + 1.5, r(K), r(K)))
    w, b = r(M, K) * K ** -0.5, r(M)
    src = S(x1, x2, coef) if two else S(x1)
    outs = []
    for mode in (0, 1):
        lib.tamgcn_set_split_mode(mode)
        y, part = ops.conv(src, K=K, w=w, bias=b, M=M, stats=True)
        outs.append((y.clone(), part.sum(2).clone(), lib.tamgcn_last_kernel().decode()))
    lib.tamgcn_set_split_mode(1)
    (y0, p0, k0), (y1, p1, k1) = outs
    ey = float((y0 - y1).abs().max() / y0.abs().max())
    ep = float((p0 - p1).abs().max() / p0.abs().max())
    worst = max(worst, ey, ep)
    print(f'N={N} K={K} M={M} T={T} V={V} two={two}: y rel err {ey:.2e}  moments rel err {ep:.2e}   [{k0}] vs [{k1}]', flush=True)
print('worst', worst)
