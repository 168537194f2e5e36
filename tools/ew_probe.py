"""GPU-box tool: achieved bandwidth of the element-wise passes at the N-UCLA and NTU layer shapes, next to torch.add
on the same tensors (what a plain streaming kernel reaches on this box at that size)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tam_gcn_amd import ops
from tam_gcn_amd.ops import S
dev = torch.device('cuda:0')


def timeit(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for shape in [(256, 64, 64, 20), (256, 256, 16, 20), (256, 64, 300, 25), (256, 128, 150, 25), (256, 256, 75, 25), (128, 256, 512, 64)]:
    N, Cc, T, V = shape
    a, b = ops.empty(*shape, like=torch.empty(1, device=dev)), ops.empty(*shape, like=torch.empty(1, device=dev))
    a.normal_(); b.normal_()
    coef = torch.randn(3, Cc, device=dev)
    nb = a.numel() * 4
    out = torch.empty_like(a)
    t = timeit(lambda: torch.add(a, b, out=out))
    print(f'{shape}: {nb/1e6:.0f} MB/tensor; torch.add (3 tensors) {3*nb/t/1e12:.2f} TB/s', end='')
    t = timeit(lambda: ops.add_act_fwd(S(a, coef=coef), S(b), True, Cc))
    print(f' | add_act_fwd (3) {3*nb/t/1e12:.2f}', end='')
    o = ops.add_act_fwd(S(a, coef=coef), S(b), True, Cc)
    t = timeit(lambda: ops.add_act_bwd(b, o, True, a, coef[0].contiguous(), None, None, True))
    print(f' | add_act_bwd (4) {4*nb/t/1e12:.2f}', end='')
    t = timeit(lambda: ops.apply(S(a, coef=coef), Cc))
    print(f' | apply (2) {2*nb/t/1e12:.2f}', end='')
    t = timeit(lambda: ops.tmean(S(a), Cc))
    print(f' | tmean (1) {nb/t/1e12:.2f}')
    del a, b, out, o
