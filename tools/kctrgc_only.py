"""ctrgc fwd / bwd at two layer shapes (or the one given as `Cin Cout T`), 2 launches each, in the training configuration
(E built once in HBM, x3 kept): target for rocprofv3 --pmc passes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tam_gcn_amd import ops
from tam_gcn_amd.ops import S
dev = torch.device('cuda:0')
N, V = 256, 20
r = lambda *s: torch.randn(*s, device=dev)
shapes = [tuple(int(v) for v in sys.argv[1:4])] if len(sys.argv) > 3 else [(64, 64, 64), (256, 256, 16)]
for Cin, Cout, T in shapes:
    R = Cin // 8
    x = r(N, Cin, T, V); pq = r(6 * R, N, V)
    W3 = r(3 * Cout, Cin) * 0.1; B3 = r(3 * Cout); W4 = r(3, Cout, R) * 0.1; B4 = r(3, Cout)
    A = r(3, V, V) * 0.1; al = torch.tensor([0.5], device=dev)
    dy = r(N, Cout, T, V); ypre = r(N, Cout, T, V); cb = r(3, Cout)
    E = ops.ctrgc_build_E(S(x), pq, W3, B3, W4, B4, A, al, Cin, Cout, 3, R)
    for _ in range(2):
        ops.ctrgc_fwd(S(x), pq, W3, B3, W4, B4, A, al, Cin, Cout, 3, R, stats=True, keep_x3=True, E=E)
        ops.ctrgc_bwd(S(x), pq, W3, B3, W4, B4, A, al, Cin, Cout, 3, R, S(dy, ypre, cb))
    gy = r(N, 3 * Cout, T, V)
    for _ in range(2):
        ops.wgrad(S(gy), S(x), M=3 * Cout, K=Cin)
torch.cuda.synchronize()
