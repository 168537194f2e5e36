"""GPU-box tool: duration of the 1x1 GEMM launches against the batch size.  A launch whose time does not fall with the batch is a
chain of latencies inside one workgroup (prologue -> operand round trips -> MFMA -> staged epilogue), not a throughput problem."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tam_gcn_amd import ops
from tam_gcn_amd.ops import S
dev = torch.device('cuda:0')


def run(N, K, M, T, two=False, bwd=False):
    x = torch.randn(N, K, T, 20, device=dev); x2 = torch.randn(N, K, T, 20, device=dev) if two else None
    coef = torch.randn(3, K, device=dev)
    if bwd:
        w = torch.randn(K, M, 1, 1, device=dev) * 0.1
        f = lambda: ops.conv(S(x, x2, coef if two else None), K=K, w=w, bias=None, M=M, wmode=1)
    else:
        w = torch.randn(M, K, 1, 1, device=dev) * 0.1; b = torch.randn(M, device=dev)
        f = lambda: ops.conv(S(x, x2, coef), K=K, w=w, bias=b, M=M, stats=True)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


for nm, K, M, T, two, bwd in (('fwd 64->64 T64', 64, 64, 64, False, False), ('fwd 256->256 T16', 256, 256, 16, False, False),
                             ('bwd 2-src 64->64 T64', 64, 64, 64, True, True), ('bwd split 768->256 T16', 768, 256, 16, False, True),
                             ('bwd split 384->128 T32', 384, 128, 32, False, True)):
    print(f'{nm:26s} ' + '  '.join(f'N={N}: {run(N, K, M, T, two, bwd):6.1f}' for N in (8, 32, 64, 128, 256, 512)) + ' us', flush=True)
