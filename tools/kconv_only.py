"""One conv shape, a few launches: target for rocprofv3 --kernel-trace / --pmc passes.
    python3 tools/kconv_only.py [--K 64 --M 64 --T 64 --two --nostats --bwd]
--bwd: the data-gradient form (transposed weight view, broadcast + residual add in the epilogue)."""
import argparse, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tam_gcn_amd import ops
from tam_gcn_amd.ops import S
ap = argparse.ArgumentParser()
ap.add_argument('--K', type=int, default=64); ap.add_argument('--M', type=int, default=64)
ap.add_argument('--T', type=int, default=64); ap.add_argument('--N', type=int, default=256)
ap.add_argument('--two', action='store_true'); ap.add_argument('--nostats', action='store_true')
ap.add_argument('--bwd', action='store_true'); ap.add_argument('--iters', type=int, default=4)
ap.add_argument('--V', type=int, default=20); ap.add_argument('--plain', action='store_true')
ap.add_argument('--kt', type=int, default=1); ap.add_argument('--dil', type=int, default=1)
a = ap.parse_args()
dev = torch.device('cuda:0')
N, V, K, M, T = a.N, a.V, a.K, a.M, a.T
x = torch.randn(N, K, T, V, device=dev); x2 = torch.randn(N, K, T, V, device=dev) if a.two else None
coef = torch.randn(3, K, device=dev)
if a.bwd:
    w = torch.randn(K, M, 1, 1, device=dev) * 0.1; a1 = torch.randn(N, M, T, V, device=dev); bc = torch.randn(M, N, V, device=dev)
    f = lambda: ops.conv(S(x, x2, coef if a.two else None), K=K, w=w, bias=None, M=M, wmode=1, bcast=bc, bcast_scale=0.1, add1=a1)
else:
    w = torch.randn(M, K, a.kt, 1, device=dev) * 0.1; b = torch.randn(M, device=dev)
    kw = dict(KT=a.kt, dil=a.dil, pad=(a.kt - 1) * a.dil // 2) if a.kt > 1 else {}
    f = lambda: ops.conv(S(x, x2, None if a.plain else coef, act=1 if a.kt > 1 else 0), K=K, w=w, bias=b, M=M, stats=not a.nostats, **kw)
for _ in range(a.iters):
    f()
torch.cuda.synchronize()
