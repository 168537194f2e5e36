"""What each part of the fused CTRGC forward costs: side builds of ctrgc.hip with -DTG_CKO=<mask> (results wrong by design) timed with
tools/kbench.py ctrgc beside the product.      here:  python tools/ctrgc_knockout.py build       box:  python tools/ctrgc_knockout.py run
masks: 1 one GEMM k4-step in eight, 2 no operand loads, 4 no stage commit, 8 one x3 tile write in nine, 16 no aggregation, 32 no copy-out stores,
64 no GEMM fragment reads."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SIDE = os.path.join(ROOT, 'tools', '_side')
MASKS = [int(m) for m in os.environ.get('CKO_MASKS', '1,2,4,8,16,32,64,6,70,71,48,127').split(',')]
if sys.argv[1] == 'build':
    from tam_gcn_amd import build as B
    B.build()
    os.makedirs(SIDE, exist_ok=True)
    csrc = os.path.join(ROOT, 'tam_gcn_amd', 'csrc')
    others = [os.path.splitext(s)[0] + '.o' for s in B.sources() if os.path.basename(s) != 'ctrgc.hip']
    for old in os.listdir(SIDE):
        os.remove(os.path.join(SIDE, old))
    procs = []
    for m in MASKS:
        o = os.path.join(SIDE, f'ctrgc_cko{m}.o')
        procs.append(subprocess.Popen([B._hipcc(), f'--offload-arch={B.ARCH}', '-O3', '-std=c++17', '-fPIC', f'-DTG_CKO={m}', '-c',
                                       os.path.join(csrc, 'ctrgc.hip'), '-o', o]))
        if len(procs) == 6:
            for p in procs: assert p.wait() == 0
            procs = []
    for p in procs: assert p.wait() == 0
    for m in MASKS:
        subprocess.check_call([B._hipcc(), f'--offload-arch={B.ARCH}', '-shared', '-fPIC', '-o', os.path.join(SIDE, f'libtamgcn_cko{m}.so'),
                               os.path.join(SIDE, f'ctrgc_cko{m}.o')] + others)
        os.remove(os.path.join(SIDE, f'ctrgc_cko{m}.o'))
    print('built', sorted(os.listdir(SIDE)))
else:
    for m in [0] + MASKS + [0]:
        env = dict(os.environ, TAMGCN_SPLIT_BF16='0')
        if m:
            env['TAMGCN_LIB'] = os.path.join(SIDE, f'libtamgcn_cko{m}.so')
        print(f'===== mask {m}', flush=True)
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'kbench.py'), 'ctrgc'], env=env, capture_output=True, text=True).stdout
        print('\n'.join(l for l in out.splitlines() if 'x3 kept' in l), flush=True)
