"""GPU-box tool: where a conv workgroup spends its time.  Builds an instrumented copy of the library
(-DTAMGCN_TRACE: s_memtime stamps of wave 0 at the phase boundaries of conv_kernel_vec), runs one shape
and prints the share of every phase.  Same arguments as tools/kconv_only.py."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
side = '/tmp/libtamgcn_trace.so'
from tam_gcn_amd import build as B
B.build(out=side, defines=('TAMGCN_TRACE',), verbose=False)
os.environ['TAMGCN_LIB'] = side
sys.argv = [sys.argv[0]] + sys.argv[1:]
import runpy, torch
from tam_gcn_amd import _lib
lib = _lib.load()
buf = (C.c_ulonglong * 16)()
lib.tamgcn_trace_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
lib.tamgcn_trace_read(buf, 1)
runpy.run_path(os.path.join(ROOT, 'tools', 'kconv_only.py'), run_name='__main__')
torch.cuda.synchronize()
lib.tamgcn_trace_read(buf, 1)
names = ['prologue (cf, first loads)', 'barrier top of chunk (reg-staged kernel)', 'wait for the chunk\'s loads (vmcnt)', 'stage A/B into LDS (reg-staged kernel)',
         'barrier', 'issue next loads', 'MFMA loop (issue)', 'epilogue', 'whole workgroup']
nb = buf[9] or 1
print(f'workgroups traced: {nb}; shader clocks per workgroup (wave 0):')
for i, nm in enumerate(names):
    print(f'  {nm:36s} {buf[i] / nb:10.0f}  {buf[i] / max(buf[8], 1):6.1%}')
