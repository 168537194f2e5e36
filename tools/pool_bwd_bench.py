"""GPU-box tool: max-pool gradient, vector kernel (default) against the LDS-staged one (TAMGCN_POOL_VEC=0), one process per arm."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'arm':
    sys.path.insert(0, ROOT)
    import torch
    from tam_gcn_amd import ops
    from tam_gcn_amd.ops import S
    dev = torch.device('cuda:0')
    for Cb, T, s in ((16, 64, 1), (32, 64, 2), (32, 32, 1), (64, 32, 2), (64, 16, 1)):
        N, V = 256, 20
        T2 = (T - 1) // s + 1
        h = torch.randn(N, 3 * Cb, T, V, device=dev); ch = torch.randn(3, 3 * Cb, device=dev)
        g1, g2 = torch.randn(N, 4 * Cb, T2, V, device=dev), torch.randn(N, 4 * Cb, T2, V, device=dev)
        cg = torch.randn(3, 4 * Cb, device=dev); mu = torch.randn(2, 3 * Cb, device=dev)
        dh = ops.empty(N, 3 * Cb, T, V, like=h)
        f = lambda: ops.maxpool_bwd(S(g1, g2, cg, coff=2 * Cb), S(h, None, ch, coff=2 * Cb, act=1), mu, Cb, s, dh, 2 * Cb)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        by = 4.0 * N * Cb * V * (2 * T2 + 2 * T)
        print(f'Cb{Cb} T{T} s{s}: {us:6.1f} us  {by / us / 1e6:5.2f} TB/s', flush=True)
else:
    for v in ('1', '0'):
        print(f'===== TAMGCN_POOL_VEC={v}', flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), 'arm'], env=dict(os.environ, TAMGCN_POOL_VEC=v))
