import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from tam_gcn_amd import ops
from tam_gcn_amd.ops import S
dev = torch.device('cuda:0')
def run(N, K, T, d=1, kt=5):
    x = torch.randn(N, K, T, 20, device=dev); coef = torch.randn(3, K, device=dev)
    w = torch.randn(K, K, kt, 1, device=dev) * 0.1; b = torch.randn(K, device=dev)
    f = lambda: ops.conv(S(x, None, coef, act=1), K=K, w=w, bias=b, M=K, stats=True, KT=kt, dil=d, pad=(kt - 1) * d // 2)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
for K, T in ((16, 64), (32, 32), (64, 16)):
    print(f'K{K} T{T}: ' + '  '.join(f'N={N}: {run(N, K, T):6.1f} us' for N in (16, 32, 64, 128, 256, 512)), flush=True)
