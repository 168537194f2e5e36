"""GPU-box tool: time the C-ABI ops at the benchmark's layer shapes (N=256, N-UCLA V=20) with HIP
events; prints microseconds, algorithmic GB/s (vs 8 TB/s) and fp32 TFLOP/s (vs 157.3)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tam_gcn_amd import ops                    # noqa: E402
from tam_gcn_amd.ops import S                  # noqa: E402

dev = torch.device('cuda:0')
N, V = 256, 20
only = sys.argv[1:] if len(sys.argv) > 1 else None


def timeit(fn, iters=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def rep(name, us, byt, fl):
    print(f'{name:44s} {us:9.1f} us  {byt / us / 1e3:8.1f} GB/s ({byt / us / 1e3 / 8000:5.1%})  '
          f'{fl / us / 1e6:7.2f} TF/s ({fl / us / 1e6 / 157.3:5.1%})', flush=True)


r = lambda *s: torch.randn(*s, device=dev)

if not only or 'conv' in only:
    for nm, K, M, T, KT, dil, st, two in [('1x1 64->64 T64 (offset)', 64, 64, 64, 1, 1, 1, True),
                                          ('1x1 64->48 T64 (tcn entry)', 64, 48, 64, 1, 1, 1, False),
                                          ('1x1 128->128 T32', 128, 128, 32, 1, 1, 1, True),
                                          ('1x1 256->256 T16', 256, 256, 16, 1, 1, 1, True),
                                          ('1x1 256->192 T16 (tcn entry)', 256, 192, 16, 1, 1, 1, False),
                                          ('tconv 16->16 k5 d2 T64', 16, 16, 64, 5, 2, 1, False),
                                          ('tconv 64->64 k5 d1 T16', 64, 64, 16, 5, 1, 1, False),
                                          ('1x1 3->64 T64 (l1 down)', 3, 64, 64, 1, 1, 1, False)]:
        x = r(N, K, T, V); x2 = r(N, K, T, V) if two else None
        coef = r(3, K)
        w = r(M, K, KT, 1) * 0.1; b = r(M)
        pad = (KT + (KT - 1) * (dil - 1) - 1) // 2
        us = timeit(lambda: ops.conv(S(x, x2, coef), K=K, w=w, bias=b, M=M, KT=KT, dil=dil, stride=st, pad=pad, stats=True))
        rep('conv fwd ' + nm, us, 4.0 * N * T * V * (K * (2 if two else 1) + M), 2.0 * N * M * K * KT * T * V)
    for nm, K, M, T in [('dx<-dx3 192->64 T64', 192, 64, 64), ('dx<-dx3 384->128 T32', 384, 128, 32),
                        ('dx<-dx3 768->256 T16', 768, 256, 16)]:
        x = r(N, K, T, V); w = r(K, M, 1, 1) * 0.1; a1 = r(N, M, T, V); bc = r(M, N, V)
        us = timeit(lambda: ops.conv(S(x), K=K, w=w, bias=None, M=M, wmode=1, bcast=bc, bcast_scale=0.1, add1=a1))
        rep('conv bwd-data ' + nm, us, 4.0 * N * T * V * (K + 2 * M), 2.0 * N * M * K * T * V)

if not only or 'wgrad' in only:
    for nm, M, K, T, KT, two in [('64x64 T64', 64, 64, 64, 1, True), ('192x64 T64 (dW3)', 192, 64, 64, 1, False),
                                 ('384x128 T32 (dW3)', 384, 128, 32, 1, False), ('768x256 T16 (dW3)', 768, 256, 16, 1, False),
                                 ('256x256 T16', 256, 256, 16, 1, True), ('k5 16x16 T64', 16, 16, 64, 5, True),
                                 ('k5 64x64 T16', 64, 64, 16, 5, True)]:
        gy = r(N, M, T, V); g2 = r(N, M, T, V) if two else None; cg = r(3, M)
        x = r(N, K, T, V); cx = r(3, K)
        pad = 2 if KT == 5 else 0
        us = timeit(lambda: ops.wgrad(S(gy, g2, cg), S(x, None, cx, act=1), M=M, K=K, KT=KT, pad=pad))
        rep('wgrad ' + nm, us, 4.0 * N * T * V * (M * (2 if two else 1) + K), 2.0 * N * M * K * KT * T * V)

if not only or 'ctrgc' in only:
    for nm, Cin, Cout, T in [('l1 3->64 T64', 3, 64, 64), ('l2 64->64 T64', 64, 64, 64), ('l5 64->128 T64', 64, 128, 64),
                             ('l6 128->128 T32', 128, 128, 32), ('l8 128->256 T32', 128, 256, 32), ('l9 256->256 T16', 256, 256, 16)]:
        R = 8 if Cin == 3 else Cin // 8
        S_ = 3
        x = r(N, Cin, T, V)
        pq = r(S_ * 2 * R, N, V)
        W3 = r(S_ * Cout, Cin) * 0.1; B3 = r(S_ * Cout); W4 = r(S_, Cout, R) * 0.1; B4 = r(S_, Cout)
        A = r(S_, V, V) * 0.1; al = torch.tensor([0.5], device=dev)
        fl = N * S_ * (2.0 * Cin * Cout * T * V + 2.0 * R * Cout * V * V + 2.0 * Cout * T * V * V)
        # the training configuration: E built once per layer, forward loads its tiles and keeps x3
        us = timeit(lambda: ops.ctrgc_build_E(S(x), pq, W3, B3, W4, B4, A, al, Cin, Cout, S_, R))
        rep('ctrgc_build_e ' + nm, us, 4.0 * N * S_ * Cout * V * V, N * S_ * 2.0 * R * Cout * V * V)
        Eg = ops.ctrgc_build_E(S(x), pq, W3, B3, W4, B4, A, al, Cin, Cout, S_, R)
        us = timeit(lambda: ops.ctrgc_fwd(S(x), pq, W3, B3, W4, B4, A, al, Cin, Cout, S_, R, stats=True, keep_x3=True, E=Eg))
        rep('ctrgc_fwd (E loaded, x3 kept; SURVEY bytes) ' + nm[:14], us, 4.0 * N * T * V * (Cin + Cout), fl)
        us = timeit(lambda: ops.ctrgc_fwd(S(x), pq, W3, B3, W4, B4, A, al, Cin, Cout, S_, R, stats=False, E=Eg))
        rep('ctrgc_fwd (E loaded, no x3 store: inference) ' + nm[:14], us, 4.0 * N * T * V * (Cin + Cout), fl)
        _, _, x3 = ops.ctrgc_fwd(S(x), pq, W3, B3, W4, B4, A, al, Cin, Cout, S_, R, stats=True, keep_x3=True, E=Eg)
        dy = r(N, Cout, T, V); ypre = r(N, Cout, T, V); cb = r(3, Cout)
        ca = (S(x), pq, W3, B3, W4, B4, A, al, Cin, Cout, S_, R, S(dy, ypre, cb))
        us = timeit(lambda: ops.ctrgc_bwd_dx3(*ca, E=Eg))
        rep('ctrgc_bwd_dx3 ' + nm, us, 4.0 * N * T * V * (2 * Cout + 3 * Cout), N * S_ * (2.0 * R * Cout * V * V + 2.0 * Cout * T * V * V))
        us = timeit(lambda: ops.ctrgc_bwd_de(*ca, x3=x3))
        rep('ctrgc_bwd_de (acc + tail) ' + nm, us, 4.0 * N * T * V * (2 * Cout + 3 * Cout), N * S_ * (2.0 * R * Cout * V * V + 2.0 * Cout * T * V * V))

if not only or 'ew' in only:
    for C_, T in [(64, 64), (256, 16)]:
        y, o, x = r(N, C_, T, V), r(N, C_, T, V), r(N, C_, T, V)
        cy, co = r(3, C_), r(3, C_)
        us = timeit(lambda: ops.gcn_tail_fwd(S(y, coef=cy), S(o, coef=co), S(x)))
        rep(f'gcn_tail_fwd C{C_} T{T}', us, 4.0 * N * C_ * T * V * 4, 0)
        sv = r(2, C_)
        us = timeit(lambda: ops.gcn_tail_bwd(y, x, S(o, coef=co), sv))
        rep(f'gcn_tail_bwd C{C_} T{T}', us, 4.0 * N * C_ * T * V * 5, 0)
