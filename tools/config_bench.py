"""GPU-box tool: fwd + CE + bwd time of the other BASELINE configurations (not bench lines; bench.py measures
configs[1]): N-UCLA at the reference's real clip length T = 52, and NTU-RGB+D (25 joints, 300 frames, 2 persons)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tam_gcn_amd.models.ctrgcn import Model
dev = torch.device('cuda:0')
cases = [('N-UCLA T=52', dict(num_class=10, num_point=20, num_person=1, graph='graph.ucla.Graph', graph_args=dict(labeling_mode='spatial')), (256, 3, 52, 20, 1)),
         ('NTU T=300 M=2', dict(num_class=60, num_point=25, num_person=2, graph='graph.ntu_rgb_d.Graph', graph_args=dict(labeling_mode='spatial')),
          (int(sys.argv[1]) if len(sys.argv) > 1 else 32, 3, 300, 25, 2))]
for name, margs, shape in cases:
    torch.manual_seed(0)
    m = Model(**margs).to(dev).train()
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.endswith('alpha'):
                p.fill_(0.5)
    x = (torch.rand(*shape, device=dev) * 2 - 1)
    lab = torch.randint(0, margs['num_class'], (shape[0],), device=dev)
    def step():
        for p in m.parameters():
            p.grad = None
        torch.nn.functional.cross_entropy(m(x), lab).backward()
    step(); step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f'{name:16s} batch {shape[0]:4d}: {dt * 1e3:8.1f} ms / step (eager launches)  {shape[0] / dt:8.1f} clips/s   peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB', flush=True)
    del m, x
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
