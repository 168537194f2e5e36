"""GPU-box tool: ONE fused MS-TCN second-stage launch shape in a loop (for rocprofv3 --pmc passes, tools/pmc_kernel.sh).
    python tools/tconv_only.py --Cb 16 --T 64 --V 20 --s 1 --N 256 [--bwd] [--reps 30]"""
import argparse, os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from tam_gcn_amd import ops
from tam_gcn_amd.ops import S
ap = argparse.ArgumentParser()
for k, v in (('Cb', 16), ('T', 64), ('V', 20), ('s', 1), ('N', 256), ('reps', 30), ('kt', 5)):
    ap.add_argument('--' + k, type=int, default=v)
ap.add_argument('--bwd', action='store_true')
a = ap.parse_args()
dev = torch.device('cuda:0')
dils = (1, 2)
nb, Cb, T, V, s, N, kt = 2, a.Cb, a.T, a.V, a.s, a.N, a.kt
Ch, T2 = 3 * Cb, (a.T - 1) // a.s + 1
h = torch.randn(N, Ch, T, V, device=dev); coef = torch.randn(3, Ch, device=dev)
ws = [torch.randn(Cb, Cb, kt, 1, device=dev) * 0.1 for _ in dils]; bs = [torch.randn(Cb, device=dev) for _ in dils]
y = ops.empty(N, Ch + Cb, T2, V, like=h)
g1, g2 = torch.randn(N, Ch + Cb, T2, V, device=dev), torch.randn(N, Ch + Cb, T2, V, device=dev)
cg = torch.randn(3, Ch + Cb, device=dev); mu = torch.randn(2, Ch, device=dev); dh = ops.empty(N, Ch, T, V, like=h)
for _ in range(a.reps):
    if a.bwd:
        ops.tconv_bwd(S(g1, g2, cg), Cb, kt, list(dils), s, ws, S(h, coef=coef), mu, dh, 0)
    else:
        ops.tconv_fwd(S(h, None, coef, act=1), Cb, kt, list(dils), s, ws, bs, True, y, 0, stats=True)
torch.cuda.synchronize()
print('done')
