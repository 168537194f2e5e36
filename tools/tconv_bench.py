"""GPU-box tool: the MS-TCN second stage per layer shape (N-UCLA, NTU, config 4): the fused launches (csrc/tconv.hip) against
the per-branch launches they replace (tamgcn_conv x nb + tamgcn_maxpool_fwd; backward: tamgcn_conv x nb), back to back on one
stream, HIP events.  Roofs: HBM 6 TB/s on the algorithmic bytes, fp32 MFMA 157.3 TFLOP/s on the flops.
    python tools/tconv_bench.py [N]"""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from tam_gcn_amd import ops
from tam_gcn_amd.ops import S
dev = torch.device('cuda:0')
N0 = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def timeit(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def run(N, Cb, T, V, s, kt=5, dils=(1, 2)):
    nb = len(dils)
    Ch = (nb + 1) * Cb
    T2 = (T - 1) // s + 1
    h = torch.randn(N, Ch, T, V, device=dev)
    coef = torch.randn(3, Ch, device=dev)
    ws = [torch.randn(Cb, Cb, kt, 1, device=dev) * 0.1 for _ in dils]
    bs = [torch.randn(Cb, device=dev) for _ in dils]
    y = ops.empty(N, Ch + Cb, T2, V, like=h)
    src = lambda b: S(h, None, coef, coff=b * Cb, act=1)

    def old_f():
        for b, d in enumerate(dils):
            ops.conv(src(b), K=Cb, w=ws[b], bias=bs[b], M=Cb, KT=kt, dil=d, stride=s, pad=(kt - 1) * d // 2, y=y, ycoff=b * Cb, T_out=T2, stats=True)
        ops.maxpool_fwd(src(nb), Cb, s, y, nb * Cb, stats=True)

    def new_f():
        ops.tconv_fwd(S(h, None, coef, act=1), Cb, kt, list(dils), s, ws, bs, True, y, 0, stats=True)
    g1, g2 = torch.randn(N, Ch + Cb, T2, V, device=dev), torch.randn(N, Ch + Cb, T2, V, device=dev)
    cg = torch.randn(3, Ch + Cb, device=dev)
    mu = torch.randn(2, Ch, device=dev)
    dh = ops.empty(N, Ch, T, V, like=h)
    gy = lambda b: S(g1, g2, cg, coff=b * Cb)

    def old_b():
        for b, d in enumerate(dils):
            pad = (kt - 1) * d // 2
            ops.conv(gy(b), K=Cb, w=ws[b], bias=None, M=Cb, KT=kt, dil=d, stride=1, pad=(kt - 1) * d - pad, wmode=1, up=s, y=dh,
                     ycoff=b * Cb, T_out=T, mask=S(h, coef=coef, coff=b * Cb), aux=h, aux_center=mu, auxcoff=b * Cb, stats=True)

    def new_b():
        ops.tconv_bwd(gy(0), Cb, kt, list(dils), s, ws, S(h, coef=coef), mu, dh, 0)
    def old_w():
        return [ops.wgrad(gy(b), S(h, coef=coef, coff=b * Cb, act=1), M=Cb, K=Cb, KT=kt, dil=d, stride=s, pad=(kt - 1) * d // 2) for b, d in enumerate(dils)]

    def new_w():
        return ops.tconv_wgrad(gy(0), S(h, coef=coef, act=1), Cb, kt, list(dils), s)
    fl = 2.0 * N * nb * Cb * Cb * kt * T2 * V
    by_f = 4.0 * N * (nb + 1) * Cb * (T + T2) * V
    by_b = 4.0 * N * nb * Cb * (2 * T2 + 2 * T) * V
    roof = lambda by: max(by / 6e12, fl / 157.3e12) * 1e6
    o_f, n_f, o_b, n_b = timeit(old_f), timeit(new_f), timeit(old_b), timeit(new_b)
    o_w, n_w = timeit(old_w), timeit(new_w)
    by_w = 4.0 * N * nb * Cb * (2 * T2 + T) * V
    print(f'Cb{Cb:3d} T{T:4d} V{V:3d} s{s} N{N:4d}: fwd old {o_f:7.1f} new {n_f:7.1f} us (roof {roof(by_f):6.1f}, {roof(by_f) / n_f:4.0%})   '
          f'bwd old {o_b:7.1f} new {n_b:7.1f} us (roof {roof(by_b):6.1f}, {roof(by_b) / n_b:4.0%})   '
          f'wgrad old {o_w:7.1f} new {n_w:7.1f} us (roof {roof(by_w):6.1f}, {roof(by_w) / n_w:4.0%})', flush=True)


for Cb, T, V, s in ((16, 64, 20, 1), (32, 64, 20, 2), (32, 32, 20, 1), (64, 32, 20, 2), (64, 16, 20, 1)):
    run(N0, Cb, T, V, s)
for Cb, T, V, s in ((16, 300, 25, 1), (32, 300, 25, 2), (32, 150, 25, 1), (64, 150, 25, 2), (64, 75, 25, 1)):
    run(N0, Cb, T, V, s)
run(max(1, N0 // 8), 64, 512, 64, 1)
