"""Build libtamgcn.so (HIP, gfx950 only) in-tree with hipcc.

    python -m tam_gcn_amd.build [--force]

The shared object is git-ignored but travels to the GPU box with the
snapshot; nothing is JIT-compiled at import time.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = ['lib.hip', 'conv.hip', 'bn.hip', 'elementwise.hip', 'ctrgc.hip', 'ctrgc_de.hip', 'ctrgc_tiled.hip', 'stemhead.hip', 'feeder.hip', 'f2.hip', 'tconv.hip']
LIB = os.path.join(HERE, 'libtamgcn.so')
ARCH = 'gfx950'


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def sources():
    return [os.path.join(HERE, 'csrc', s) for s in SRC]


def needs_build():
    if not os.path.exists(LIB):
        return True
    deps = sources() + [os.path.join(HERE, 'csrc', 'common.h'),
                        os.path.join(HERE, '..', 'include', 'tamgcn.h')]
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, out=None, defines=()):
    """out/defines: instrumented side builds for tools/ (e.g. -DTAMGCN_TRACE); the product is LIB."""
    lib = out or LIB
    if out is None and not force and not needs_build():
        return LIB
    objs = []
    procs = []
    for s in sources():
        o = os.path.splitext(s)[0] + ('.o' if out is None else '.side.o')
        objs.append(o)
        cmd = [_hipcc(), f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', *[f'-D{d}' for d in defines], '-c', s, '-o', o]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd) + '\n' + out.decode(errors='replace'))
    cmd = [_hipcc(), f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', lib] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode:
        raise RuntimeError('link failed: ' + ' '.join(cmd) + '\n' + r.stdout.decode(errors='replace'))
    if verbose:
        print(f'[tam_gcn_amd.build] built {lib}')
    return lib


if __name__ == '__main__':
    build(force='--force' in sys.argv)
