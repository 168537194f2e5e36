"""What each part of the 1x1 LDS-DMA GEMM costs: side builds of conv.hip with -DTG_KO=<mask> (results wrong by design) timed at
the step's launch signatures beside the product.   here:  python tools/conv_knockout.py build      (hipcc, no GPU needed)
                                                    box:   python tools/conv_knockout.py run
masks: 1 no epilogue, 2 no prologue arithmetic on the B fragments, 4 no A DMA, 8 no B DMA.  (Round 4's first run also had 16 no MFMAs and
32 no fragment reads, on the kernel as it was then: profiles/r04_conv_knockout.txt.)  The run also times the 4-wave layout (TAMGCN_CONV_WAVES=4)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SIDE = os.path.join(ROOT, 'tools', '_side')
MASKS = [0, 1, 13, 15]
EXTRA = {}
if sys.argv[1] == 'build':
    from tam_gcn_amd import build as B
    B.build()
    os.makedirs(SIDE, exist_ok=True)
    csrc = os.path.join(ROOT, 'tam_gcn_amd', 'csrc')
    others = [os.path.splitext(s)[0] + '.o' for s in B.sources() if os.path.basename(s) != 'conv.hip']
    procs = []
    variants = {f'ko{m}': [f'-DTG_KO={m}'] for m in MASKS[1:]}
    variants.update(EXTRA)
    for old in os.listdir(SIDE):
        os.remove(os.path.join(SIDE, old))
    for nm, defs in variants.items():
        o = os.path.join(SIDE, f'conv_{nm}.o')
        procs.append(subprocess.Popen([B._hipcc(), f'--offload-arch={B.ARCH}', '-O3', '-std=c++17', '-fPIC', *defs, '-c',
                                       os.path.join(csrc, 'conv.hip'), '-o', o]))
        if len(procs) == 4:
            for p in procs: assert p.wait() == 0
            procs = []
    for p in procs: assert p.wait() == 0
    for nm in variants:
        subprocess.check_call([B._hipcc(), f'--offload-arch={B.ARCH}', '-shared', '-fPIC', '-o', os.path.join(SIDE, f'libtamgcn_{nm}.so'),
                               os.path.join(SIDE, f'conv_{nm}.o')] + others)
        os.remove(os.path.join(SIDE, f'conv_{nm}.o'))
    print('built', sorted(os.listdir(SIDE)))
else:
    for nm in ['product', 'waves4', 'product', 'waves4'] + [f'ko{m}' for m in MASKS[1:]] + [f'ko{m}_waves4' for m in MASKS[1:]]:
        env = dict(os.environ, TAMGCN_SPLIT_BF16='0')
        if nm.endswith('waves4'):
            env['TAMGCN_CONV_WAVES'] = '4'
        if nm.startswith('ko'):
            env['TAMGCN_LIB'] = os.path.join(SIDE, f'libtamgcn_{nm.split("_")[0]}.so')
        print(f'===== {nm}', flush=True)
        subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'conv_swap_bench.py'), 'arm'], env=env)
