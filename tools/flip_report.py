"""GPU-box diagnostic: is an end-to-end gradient deviation of a model case a ReLU / max-pool decision that falls on the other
side in fp32, or an error?  (1) the fp64 oracle forward with every ReLU input and every max-pool window monitored: how close to
a tie the closest decisions are, and in which block; (2) the HIP model twice -- temporal branches on csrc/tconv.hip and on the
generic kernels (ops.TCONV) -- with every block's output and input gradient recorded: the block where the two runs' gradients
part, and the forward difference of the two runs at that block (a flip shows as a forward difference of ~1e-7 with a gradient
difference of ~1e-2 downstream of ONE block).     python tools/flip_report.py [tag]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from cases import MODEL_CASES, MODEL_PARAM_SEED, MODEL_X_SEED, MODEL_LABEL_SEED   # noqa: E402
from params import fill_state_, make_input, make_labels                           # noqa: E402
from tam_gcn_amd import ops                                                         # noqa: E402
from tam_gcn_amd.models import ctrgcn as M                                          # noqa: E402
from oracle import ctrgcn_oracle as O                                               # noqa: E402
tagsel = sys.argv[1] if len(sys.argv) > 1 else 'ucla_t13'
tag, margs, shape = next(c for c in MODEL_CASES if c[0] == tagsel)
dev = torch.device('cuda:0')
m0 = M.Model(**margs)
fill_state_(m0.state_dict(), seed=MODEL_PARAM_SEED)
sd = {k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in m0.state_dict().items()}
x = make_input(shape, seed=MODEL_X_SEED)
lab = make_labels(shape[0], margs['num_class'], seed=MODEL_LABEL_SEED)

# ---- (1) fp64 oracle, decisions monitored per block
relu0, pool0 = torch.relu, torch.nn.functional.max_pool2d
cur = {'blk': 0}
near = {}


def relu(t):
    a = t.detach().abs()
    near.setdefault(cur['blk'], []).append(('relu', float(a.min()), int((a < 1e-6).sum()), int((a < 1e-5).sum()), a.numel()))
    return relu0(t)


def pool(t, *a, **k):
    d = t.detach()
    p = torch.nn.functional.pad(d, (0, 0, 1, 1), value=float('-inf'))
    w = torch.stack((p[:, :, :-2], p[:, :, 1:-1], p[:, :, 2:]), 0)
    s = w.sort(0).values
    gap = (s[2] - s[1])[..., ::1]
    gap = gap[torch.isfinite(gap)]
    near.setdefault(cur['blk'], []).append(('pool', float(gap.min()), int((gap < 1e-6).sum()), int((gap < 1e-5).sum()), gap.numel()))
    return pool0(t, *a, **k)


torch.relu, torch.nn.functional.max_pool2d = relu, pool
h, N, Mp = O._stem(x.double(), sd, margs['num_point'], True)
for i in range(1, 11):
    cur['blk'] = i
    h = O.tcn_gcn_unit(h, sd, f'l{i}', O._STRIDES.get(i, 1), residual=(i != 1), training=True)
torch.relu, torch.nn.functional.max_pool2d = relu0, pool0
print(f'{tag}: fp64 oracle forward, closest decisions per block (kind: min distance to a tie, # within 1e-6, # within 1e-5 of #)')
for b in sorted(near):
    print(f'  l{b}: ' + ' | '.join(f'{k} {mn:.2e} {c6} {c5} /{n}' for k, mn, c6, c5, n in near[b]))


# ---- (2) the HIP model with the temporal branches on tconv.hip and on the generic kernels
def run(tconv):
    ops.TCONV = tconv
    m = M.Model(**margs)
    fill_state_(m.state_dict(), seed=MODEL_PARAM_SEED)
    m = m.to(dev).train()
    outs, gins = {}, {}
    hooks = []
    for i in range(1, 11):
        blk = getattr(m, f'l{i}')
        hooks.append(blk.register_forward_hook(lambda mod, inp, out, i=i: outs.__setitem__(i, (out[0] if isinstance(out, tuple) else out).detach().cpu().double())))
        hooks.append(blk.register_full_backward_hook(lambda mod, gi, go, i=i: gins.__setitem__(i, gi[0].detach().cpu().double()) if gi[0] is not None else None))
    xg = x.to(dev).requires_grad_(True)
    torch.nn.functional.cross_entropy(m(xg), lab.to(dev)).backward()
    torch.cuda.synchronize()
    gins[0] = xg.grad.detach().cpu().double()
    return outs, gins


oa, ga = run(True)
ob, gb = run(False)
print('block: forward |tconv - generic| / max, masks differing;  input-gradient |tconv - generic| / max')
for i in range(1, 11):
    fo = float((oa[i] - ob[i]).abs().max() / ob[i].abs().max())
    mk = int(((oa[i] > 0) != (ob[i] > 0)).sum())
    g = f'{float((ga[i] - gb[i]).abs().max() / gb[i].abs().max()):.2e}' if i in ga and i in gb else '   -   '
    print(f'  l{i:<2d}  fwd {fo:.2e}  masks {mk:3d}   d(input) {g}')
print(f'  dx   {float((ga[0] - gb[0]).abs().max() / gb[0].abs().max()):.2e}')
