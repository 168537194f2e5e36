"""GPU-box tool: where a temporal workgroup of csrc/tconv.hip spends its time.  Builds an instrumented copy of the library
(-DTAMGCN_TRACE: s_memtime stamps of thread 0 of every temporal workgroup), runs tools/tconv_only.py with the given arguments
and prints shader clocks per phase.  (The trace build drains the loads explicitly before the staging pass, so that
"waiting for the next item's loads" is its own row.)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
side = '/tmp/libtamgcn_trace.so'
from tam_gcn_amd import build as B
B.build(out=side, defines=('TAMGCN_TRACE',), verbose=False)
os.environ['TAMGCN_LIB'] = side
import runpy, torch
from tam_gcn_amd import _lib
lib = _lib.load()
buf = (C.c_ulonglong * 16)()
lib.tamgcn_trace_read_tconv.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
lib.tamgcn_trace_read_tconv(buf, 1)
runpy.run_path(os.path.join(ROOT, 'tools', 'tconv_only.py'), run_name='__main__')
torch.cuda.synchronize()
lib.tamgcn_trace_read_tconv(buf, 1)
names = ['prologue: first loads -> image 0 staged', 'issue the next item\'s loads', 'MFMA loop', 'wait for the next item\'s loads (vmcnt 0)',
         'stage the next item into LDS', 'epilogue (stores issued)', 'barrier', '(items)', 'whole workgroup']
nwg, nit = buf[9] or 1, buf[7] or 1
print(f'workgroups traced: {nwg}, items per workgroup {nit / nwg:.1f}; shader clocks per workgroup (thread 0):')
for i, nm in enumerate(names):
    if i == 7:
        continue
    print(f'  {nm:44s} {buf[i] / nwg:10.0f}  {buf[i] / max(buf[8], 1):6.1%}' + (f'   ({buf[i] / nit:8.0f} per item)' if 0 < i < 7 else ''))
