"""GPU-box experiment: forward tconv launch time for two shapes (HIP events), optionally without the pooled branch."""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from tam_gcn_amd import ops
from tam_gcn_amd.ops import S
dev = torch.device('cuda:0')
def t(f, reps=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for Cb, T in ((16, 64), (32, 32), (64, 16)):
    N, V, kt, dils = 256, 20, 5, (1, 2)
    h = torch.randn(N, 3 * Cb, T, V, device=dev); coef = torch.randn(3, 3 * Cb, device=dev)
    ws = [torch.randn(Cb, Cb, kt, 1, device=dev) * 0.1 for _ in dils]; bs = [torch.randn(Cb, device=dev) for _ in dils]
    y = ops.empty(N, 4 * Cb, T, V, like=h)
    res = []
    for pool in (True, False):
        for st in (True, False):
            res.append(t(lambda: ops.tconv_fwd(S(h, None, coef, act=1), Cb, kt, list(dils), 1, ws, bs, pool, y, 0, stats=st)))
    print(f'Cb{Cb} T{T}: pool+stats {res[0]:.1f}  pool {res[1]:.1f}  nopool+stats {res[2]:.1f}  nopool {res[3]:.1f} us', flush=True)
