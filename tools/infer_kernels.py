"""GPU-box tool: per-kernel time of ONE eval-mode forward at a small batch (HIP events around every ABI launch, eager)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
from tam_gcn_amd import _lib, functional as Fn
from tam_gcn_amd.models.ctrgcn import Model
probe = B._Probe(_lib.load()); _lib._lib = probe
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = Model(**B.MODEL_ARGS); B.dedegenerate_(m); m = m.to(dev).eval()
Bn = int(sys.argv[1]) if len(sys.argv) > 1 else 1
x = torch.rand(Bn, 3, 64, 20, 1, device=dev) * 2 - 1
Fn.USE_SIDE_STREAMS = False
with torch.no_grad():
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    agg, _ = B.instrumented_pass(lambda: m(x), probe, 5)
tot = sum(a['ms'] for a in agg.values()) / 5
print(f'batch {Bn}: {sum(a["calls"] for a in agg.values()) // 5} launches, kernel time {tot * 1e3:.0f} us per forward')
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]['ms']):
    print(f'  {a["ms"] / 5 * 1e3:8.1f} us  {a["calls"] // 5:3d} x {a["ms"] / a["calls"] * 1e3:7.1f} us  {k}')
