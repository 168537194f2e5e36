"""One conv shape, a few launches: target for rocprofv3 --pmc passes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tam_gcn_amd import ops
from tam_gcn_amd.ops import S
dev = torch.device('cuda:0')
N, V, K, M, T = 256, 20, 64, 64, 64
two = len(sys.argv) > 1 and sys.argv[1] == 'two'
x = torch.randn(N, K, T, V, device=dev); x2 = torch.randn(N, K, T, V, device=dev) if two else None
coef = torch.randn(3, K, device=dev); w = torch.randn(M, K, 1, 1, device=dev) * 0.1; b = torch.randn(M, device=dev)
stats = not (len(sys.argv) > 2 and sys.argv[2] == 'nostats')
for _ in range(4):
    ops.conv(S(x, x2, coef), K=K, w=w, bias=b, M=M, stats=stats)
torch.cuda.synchronize()
