"""GPU-box tool: every ABI launch signature of one N-UCLA training step (256 clips) with its measured duration against
the two roofs of its algorithmic bytes / flops (bench.py's model), sorted by the time it loses to its binding roof.
    python tools/roofline_table.py [--side 0|1] > gpurun_out/roofline_table.txt
--side 0 runs everything on one stream (durations without co-running kernels)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import argparse
import torch
import bench
from tam_gcn_amd import _lib, functional as Fn
from tam_gcn_amd.models.ctrgcn import Model

ap = argparse.ArgumentParser()
ap.add_argument('--side', type=int, default=0)
ap.add_argument('--batch', type=int, default=0)
ap.add_argument('--config', default='ucla', choices=['ucla', 'ntu'])
a = ap.parse_args()
Fn.USE_SIDE_STREAMS = bool(a.side)
HBM, MF = 6.0e12, bench.F32_MFMA_PEAK        # 6 TB/s: what a streaming kernel reaches on this box


def sig(name, args):
    try:
        d = args[0]._obj
    except AttributeError:
        return name
    if name == 'tamgcn_conv':
        ex = (1 if d.add1 else 0) + (1 if d.add2 else 0) + (1 if d.mask else 0) + (1 if d.aux else 0)
        return f'conv K{d.K} M{d.M} KT{d.KT} T{d.T_in}>{d.T_out} src{bench._src_reads(d.src)} ex{ex} wm{d.wmode} st{1 if d.stats_part else 0} s{d.stride} N{d.N}' + (f' [al{(d.src.x1 or 0) % 16}/{(d.src.x2 or 0) % 16}/{(d.w or 0) % 16} coff{d.src.coff}/{d.src.ctot} ycoff{d.ycoff}/{d.yctot} os{d.ostride} up{d.up} bc{1 if d.bcast else 0}]' if os.environ.get('RT_DETAIL') else '')
    if name == 'tamgcn_wgrad':
        return f'wgrad K{d.K} M{d.M} KT{d.KT} T{d.T_in}>{d.T_out} gy{bench._src_reads(d.gy)} src{bench._src_reads(d.src)} N{d.N}'
    if name in ('tamgcn_tconv_fwd', 'tamgcn_tconv_bwd', 'tamgcn_tconv_wgrad'):
        return f'{name[7:]} Cb{d.Cb} nb{d.nb} KT{d.KT} T{d.T_in}>{d.T_out} s{d.stride}' + (' +pool' if name == 'tamgcn_tconv_fwd' and d.pool else '') + f' N{d.N}'
    if name.startswith('tamgcn_ctrgc'):
        return f'{name[7:]} {d.Cin}>{d.Cout} T{d.T}'
    if name in ('tamgcn_add_act_fwd', 'tamgcn_add_act_bwd', 'tamgcn_gcn_tail_fwd', 'tamgcn_gcn_tail_bwd', 'tamgcn_gcn_mid_bwd'):
        ints = [x for x in args if isinstance(x, int)]
        return f'{name[7:]} ' + 'x'.join(str(v) for v in ints[-4:])
    return name


class P(bench._Probe):
    def __getattr__(self, name):
        fn = bench._Probe.__getattr__(self, name)
        raw = getattr(self._lib, name)
        if fn is raw:
            return fn

        def wrapped(*args):
            if not self.on:
                return raw(*args)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); rc = raw(*args); e1.record()
            self.records.append((sig(name, args), self._lib.tamgcn_last_kernel().decode(), e0, e1, bench._algorithmic(name, args)))
            return rc
        return wrapped


dev = torch.device('cuda:0')
probe = P(_lib.load())
_lib._lib = probe
torch.manual_seed(0)
if a.config == 'ntu':
    margs = dict(num_class=60, num_point=25, num_person=2, graph='graph.ntu_rgb_d.Graph', graph_args=dict(labeling_mode='spatial'))
    shape = (a.batch or 128, 3, 300, 25, 2)
else:
    margs, shape = bench.MODEL_ARGS, (a.batch or 256, 3, bench.T_FRAMES, bench.V_JOINTS, 1)
m = Model(**margs)
bench.dedegenerate_(m)
m = m.to(dev).train()
x = (torch.rand(*shape) * 2 - 1).to(dev)
lab = torch.randint(0, margs['num_class'], (shape[0],)).to(dev)
ce = Fn.CrossEntropyLoss()


def step():
    for p in m.parameters():
        p.grad = None
    ce(m(x), lab).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
probe.on = True
STEPS = 3
for _ in range(STEPS):
    step()
torch.cuda.synchronize()
probe.on = False
tab = {}
for s, sym, e0, e1, (b, f) in probe.records:
    t = tab.setdefault((s, sym), [0, 0.0, b, f])
    t[0] += 1; t[1] += e0.elapsed_time(e1)
rows = []
tot = 0.0
for (s, sym), (n, ms, b, f) in tab.items():
    avg = ms / n * 1e3
    roof = max(b / HBM, f / MF) * 1e6
    per_step = ms / STEPS
    tot += per_step
    rows.append((per_step - roof * n / STEPS * 1e-3 if roof else 0.0, per_step, n / STEPS, avg, roof, b / 1e6, f / 1e9, s, sym))
rows.sort(reverse=True)
print(f'side streams {a.side}; total ABI time {tot:.2f} ms / step; roofs: HBM {HBM/1e12:.1f} TB/s, fp32 MFMA {MF/1e12:.1f} TFLOP/s')
print(f'{"lost ms":>8} {"ms/step":>8} {"n/step":>6} {"avg us":>8} {"roof us":>8} {"MB":>8} {"GFLOP":>7}  signature | kernel')
for r in rows:
    print(f'{r[0]:8.3f} {r[1]:8.3f} {r[2]:6.1f} {r[3]:8.1f} {r[4]:8.1f} {r[5]:8.1f} {r[6]:7.2f}  {r[7]} | {r[8][:60]}')
