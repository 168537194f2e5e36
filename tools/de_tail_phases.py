"""GPU-box tool: phase shares of ctrgc_de_tail_kernel (instrumented side build, -DTAMGCN_TRACE).
    python tools/de_tail_phases.py [Cout R]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
side = '/tmp/libtamgcn_trace.so'
from tam_gcn_amd import build as B
B.build(out=side, defines=('TAMGCN_TRACE',), verbose=False)
os.environ['TAMGCN_LIB'] = side
import torch
from tam_gcn_amd import _lib
lib = _lib.load()
buf = (C.c_ulonglong * 16)()
lib.tamgcn_trace_read_de.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
Cout, R = (int(v) for v in (sys.argv[1:3] if len(sys.argv) > 2 else (256, 32)))
dev = torch.device('cuda:0')
N, V, S_ = 256, 20, 3
r = lambda *s: torch.randn(*s, device=dev)
dE = r(N, S_, Cout, V, V); pq = r(S_ * 2 * R, N, V); W4 = r(S_, Cout, R) * 0.1; B4 = r(S_, Cout); al = torch.tensor([0.5], device=dev)
dA = r(N, S_, V, V); dw4 = r(N, S_, Cout, R); db4 = r(N, S_, Cout); dal = r(N * S_, 1); dpq = r(1, S_ * 2 * R, N, V)
d = _lib.CtrgcDesc(N=N, Cin=Cout, Cout=Cout, S=S_, R=R, T=16, V=V, pq=pq.data_ptr(), w4=W4.data_ptr(), b4=B4.data_ptr(), alpha=al.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
f = lambda: _lib.check(lib.tamgcn_ctrgc_bwd_de_tail(C.byref(d), dE.data_ptr(), dA.data_ptr(), dw4.data_ptr(), db4.data_ptr(), dal.data_ptr(), dpq.data_ptr(), 1, st), 'tail')
f(); torch.cuda.synchronize()
lib.tamgcn_trace_read_de(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    f()
e1.record(); torch.cuda.synchronize()
lib.tamgcn_trace_read_de(buf, 1)
names = ['prologue (pq, D fill)', 'wait for the chunk (vmcnt)', 'barrier', 'issue next + flush partials', 'dG MFMA', 'dA', 'db4', 'dW4 MFMA + partial write',
         'whole workgroup', None, 'barrier + flush (single buffer) / fragment copy', 'epilogue (dS, dp / dq)']
nb = buf[9] or 1
print(f'ctrgc_de_tail Cout={Cout} R={R}: {e0.elapsed_time(e1) / 3 * 1e3:.1f} us per launch (traced build); workgroups traced {nb}; shader clocks per workgroup (wave 0):')
for i, nm in enumerate(names):
    if nm:
        print(f'  {nm:50s} {buf[i] / nb:10.0f}  {buf[i] / max(buf[8], 1):6.1%}')
