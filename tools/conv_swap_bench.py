"""GPU-box tool: the 1x1 LDS-DMA GEMM at the step's launch signatures (`arm`); round 4 used it for the register epilogue (TAMGCN_CONV_SWAP, since removed) against
the LDS-staged epilogue (TAMGCN_CONV_SWAP=0) at the step's launch signatures (256 clips, V = 20).  One process per arm
(the switch is read once)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'arm':
    sys.path.insert(0, ROOT)
    import torch
    from tam_gcn_amd import ops
    from tam_gcn_amd.ops import S
    dev = torch.device('cuda:0')
    r = lambda *s: torch.randn(*s, device=dev)

    def t(f, reps=20):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    N, V = int(os.environ.get('BENCH_N', 256)), 20
    # name, K, M, T, two sources, stats (forward) / data gradient with adds + broadcast
    for nm, K, M, T, two, kind in [('fwd 64->64 stats', 64, 64, 64, False, 'f'), ('fwd 64->48 stats', 64, 48, 64, False, 'f'),
                                   ('fwd 128->128 stats', 128, 128, 32, False, 'f'), ('fwd 256->256 stats', 256, 256, 16, False, 'f'),
                                   ('fwd 256->192 stats', 256, 192, 16, False, 'f'),
                                   ('dgrad 64<-64 two-src', 64, 64, 64, True, 'd'), ('dgrad 128<-128 two-src', 128, 128, 32, True, 'd'),
                                   ('dgrad 256<-256 two-src', 256, 256, 16, True, 'd'),
                                   ('dx 64<-192 bcast+2 adds', 192, 64, 64, False, 'x'), ('dx 128<-384 bcast+2 adds', 384, 128, 32, False, 'x'),
                                   ('dx 256<-768 bcast+2 adds', 768, 256, 16, False, 'x')]:
        x = r(N, K, T, V); x2 = r(N, K, T, V) if two else None
        coef = r(3, K)
        if kind == 'f':
            w, b = r(M, K, 1, 1) * 0.1, r(M)
            f = lambda: ops.conv(S(x, None, coef, act=1), K=K, w=w, bias=b, M=M, stats=True)
        elif kind == 'd':
            w = r(K, M, 1, 1) * 0.1
            f = lambda: ops.conv(S(x, x2, coef), K=K, w=w, bias=None, M=M, wmode=1)
        else:
            w = r(K, M, 1, 1) * 0.1
            a1, a2, bc = r(N, M, T, V), r(N, M, T, V), r(M, N, V)
            f = lambda: ops.conv(S(x), K=K, w=w, bias=None, M=M, wmode=1, bcast=bc, bcast_scale=1.0 / T, add1=a1, add2=a2)
        us = t(f)
        fl = 2.0 * N * M * K * T * V
        print(f'{nm:28s} {us:8.1f} us  {fl / us / 1e6:6.1f} TF/s', flush=True)
else:
    for sw in ('1', '0', '1', '0'):
        print(f'===== TAMGCN_CONV_SWAP={sw}', flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), 'arm'], env=dict(os.environ, TAMGCN_CONV_SWAP=sw, TAMGCN_SPLIT_BF16='0'))
