"""GPU-box tool: forward-only throughput (eval mode, no_grad) of the N-UCLA model at a few batch sizes -- what the
inference-only callers (cross-modal attention, ensemble eval, visualisation) see.  Eager launches and HIP-graph replay.
    python tools/infer_bench.py [--t T] [batch ...]        (default T = 64, batches 1 16 256)
Batches of at most TAMGCN_F2_MAX_CLIPS clips take the small-batch kernel family (tam_gcn_amd/f2.py); TAMGCN_F2=0 puts them on
the general eval path for comparison."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tam_gcn_amd.models.ctrgcn import Model
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = Model(num_class=10, num_point=20, num_person=1, graph='graph.ucla.Graph', graph_args=dict(labeling_mode='spatial')).to(dev).eval()
with torch.no_grad():
    for k, p in m.named_parameters():
        if k.endswith('alpha'):
            p.fill_(0.5)
args = sys.argv[1:]
T = 64
if args and args[0] == '--t':
    T = int(args[1]); args = args[2:]
print(f'T = {T}, TAMGCN_F2 = {os.environ.get("TAMGCN_F2", "1")}', flush=True)
for B in ([int(v) for v in args] or (1, 16, 256)):
    x = torch.rand(B, 3, T, 20, 1, device=dev) * 2 - 1
    with torch.no_grad():
        for _ in range(3):
            y = m(x)
        torch.cuda.synchronize()
        n = 50
        t0 = time.perf_counter()
        for _ in range(n):
            y = m(x)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / n
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            y = m(x)
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = m(x)
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / n
        extra = ''
        for sp in ([int(v) for v in os.environ.get('INFER_SPLITS', '').split(',') if v] if B >= 16 else []):
            from tam_gcn_amd.inference import GraphedForward
            fast = GraphedForward(m, split=sp)
            ys = fast(x); torch.cuda.synchronize()
            assert float((ys - y).abs().max()) <= 1e-4 * float(y.abs().max()), float((ys - y).abs().max())   # (slices of <= 32 clips take the f2 kernels)
            t0 = time.perf_counter()
            for _ in range(n):
                fast(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            extra += f'   split {sp}: {dt * 1e3:6.2f} ms ({B / dt:8.0f} clips/s)'
    print(f'batch {B:4d}: eager {eager * 1e3:7.2f} ms ({B / eager:9.0f} clips/s)   hip graph {graph * 1e3:7.2f} ms ({B / graph:9.0f} clips/s)' + extra, flush=True)
