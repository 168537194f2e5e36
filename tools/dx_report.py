"""GPU-box diagnostic: how the input gradient of a model case differs from the fp64 reference --
max error, where it sits, how many entries exceed the strict bound (a ReLU-mask flip shows up as a
handful of entries in one sample).  Usage: python tools/dx_report.py [tag]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from cases import MODEL_CASES, MODEL_PARAM_SEED, MODEL_X_SEED, MODEL_LABEL_SEED   # noqa: E402
from params import fill_state_, make_input, make_labels                           # noqa: E402
from tam_gcn_amd.models import ctrgcn as M                                          # noqa: E402

gold = np.load(os.path.join(ROOT, 'tests', 'golden', 'models.npz'))
want = sys.argv[1:] or ['ucla_t13']
for tag, margs, shape in MODEL_CASES:
    if tag not in want:
        continue
    dev = torch.device('cuda:0')
    m = M.Model(**margs)
    fill_state_(m.state_dict(), seed=MODEL_PARAM_SEED)
    m = m.to(dev).train()
    x = make_input(shape, seed=MODEL_X_SEED).to(dev).requires_grad_(True)
    lab = make_labels(shape[0], margs['num_class'], seed=MODEL_LABEL_SEED).to(dev)
    torch.nn.functional.cross_entropy(m(x), lab).backward()
    torch.cuda.synchronize()
    got = x.grad.cpu().numpy().astype(np.float64)
    r32, r64 = gold[f'{tag}/dx'].astype(np.float64), gold[f'{tag}/dx64']
    scale = np.abs(r64).max()
    d, dref = np.abs(got - r64), np.abs(r32 - r64)
    tol = 2e-3 * scale + 10 * dref.max()
    print(f'{tag}: dx shape {got.shape} scale {scale:.3e}  ours max err {d.max():.3e}  reference fp32 max err {dref.max():.3e}  bound {tol:.3e}')
    print(f'  entries above the bound: {(d > tol).sum()} of {d.size};  above bound/4: {(d > tol / 4).sum()};  rel-L2 {np.sqrt((d**2).sum() / (r64**2).sum()):.3e}')
    idx = np.unravel_index(np.argsort(d, axis=None)[::-1][:8], d.shape)
    for k in range(8):
        i = tuple(int(a[k]) for a in idx)
        print(f'    {i}: got {got[i]:+.5e} ref64 {r64[i]:+.5e} ref32 {r32[i]:+.5e}')
    per_n = d.reshape(d.shape[0], -1).max(1)
    print('  max err per sample:', ' '.join(f'{v:.2e}' for v in per_n))
