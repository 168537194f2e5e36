"""GPU-box tool: A/B of a compile-time variant.  Builds a side library with the given -D defines and runs a command once
per arm, interleaved, `rounds` times:   python tools/ab_side_build.py TAMGCN_OLD_PROLOGUE 2 -- python tools/kbench.py conv"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tam_gcn_amd import build as B          # noqa: E402

i = sys.argv.index('--')
defines, rounds, cmd = sys.argv[1].split(','), int(sys.argv[2]), sys.argv[i + 1:]
side = '/tmp/libtamgcn_ab.so'
B.build(out=side, defines=tuple(defines), verbose=False)
for r in range(rounds):
    for arm, env in (('side build -D' + ','.join(defines), dict(os.environ, TAMGCN_LIB=side)), ('product build', dict(os.environ))):
        print(f'===== round {r}: {arm}', flush=True)
        subprocess.run(cmd, env=env, check=False)
