"""GPU-box tool: phase shares of ctrgc_fwd_kernel (instrumented side build, -DTAMGCN_TRACE).
    python tools/ctrgc_phases.py [Cin Cout T]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
side = '/tmp/libtamgcn_trace.so'
from tam_gcn_amd import build as B
B.build(out=side, defines=('TAMGCN_TRACE',), verbose=False)
os.environ['TAMGCN_LIB'] = side
import torch
from tam_gcn_amd import _lib, ops
from tam_gcn_amd.ops import S
lib = _lib.load()
buf = (C.c_ulonglong * 16)()
lib.tamgcn_trace_read_ctrgc.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
Cin, Cout, T = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 64, 64)))
dev = torch.device('cuda:0')
N, V, S_ = 256, 20, 3
R = 8 if Cin in (3, 9) else Cin // 8
r = lambda *s: torch.randn(*s, device=dev)
x = r(N, Cin, T, V); pq = r(S_ * 2 * R, N, V)
W3 = r(S_ * Cout, Cin) * 0.1; B3 = r(S_ * Cout); W4 = r(S_, Cout, R) * 0.1; B4 = r(S_, Cout)
A = r(S_, V, V) * 0.1; al = torch.tensor([0.5], device=dev)
E = ops.ctrgc_build_E(S(x), pq, W3, B3, W4, B4, A, al, Cin, Cout, S_, R)       # the training configuration: E once per layer in HBM
f = lambda: ops.ctrgc_fwd(S(x), pq, W3, B3, W4, B4, A, al, Cin, Cout, S_, R, stats=True, keep_x3=True, E=E)
f(); torch.cuda.synchronize()
lib.tamgcn_trace_read_ctrgc(buf, 1)
for _ in range(3):
    f()
torch.cuda.synchronize()
lib.tamgcn_trace_read_ctrgc(buf, 1)
names = ['load / build E', 'x3 GEMM chunk (stage + MFMA + tile write)', 'aggregate + z -> LDS', 'barrier', 'copy-out (y, x3) + stats']
nb = buf[9] or 1
print(f'ctrgc_fwd Cin={Cin} Cout={Cout} T={T}: workgroups traced {nb}; shader clocks per workgroup (wave 0):')
for i, nm in enumerate(names):
    print(f'  {nm:46s} {buf[i] / nb:10.0f}  {buf[i] / max(buf[8], 1):6.1%}')
print(f'  {"whole workgroup":46s} {buf[8] / nb:10.0f}')
