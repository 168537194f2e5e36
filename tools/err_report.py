"""GPU-box diagnostic: per-layer dPA error of the HIP path against the fp64 reference values in
the fixture, next to the reference's own fp32-vs-fp64 noise.  Usage: python tools/err_report.py [tag]"""
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
want = sys.argv[1:] or [c[0] for c in MODEL_CASES]
for tag, margs, shape in MODEL_CASES:
    if tag not in want:
        continue
    dev = torch.device('cuda:0')
    m = M.Model(**margs)
    fill_state_(m.state_dict(), seed=MODEL_PARAM_SEED)
    m = m.to(dev).train()
    x = make_input(shape, seed=MODEL_X_SEED).to(dev).requires_grad_(True)
    lab = make_labels(shape[0], margs['num_class'], seed=MODEL_LABEL_SEED).to(dev)
    lg = m(x)
    torch.nn.functional.cross_entropy(lg, lab).backward()
    torch.cuda.synchronize()
    r64 = gold[f'{tag}/logits_train64']
    print(f'== {tag}: logits err {np.abs(lg.detach().cpu().numpy() - r64).max():.2e} '
          f'(ref noise {np.abs(gold[f"{tag}/logits_train"] - r64).max():.2e})')
    d64 = gold[f'{tag}/dx64']
    print(f'   dx   rel err {np.abs(x.grad.cpu().numpy() - d64).max() / np.abs(d64).max():.2e} '
          f'(ref noise {np.abs(gold[f"{tag}/dx"] - d64).max() / np.abs(d64).max():.2e})')
    for i in range(10, 0, -1):
        for nm in ('PA', 'alpha'):
            k = f'l{i}.gcn1.{nm}'
            g = dict(m.named_parameters())[k].grad.cpu().numpy()
            a, b = gold[f'{tag}/grad/{k}'], gold[f'{tag}/grad64/{k}']
            sc = np.abs(b).max() + 1e-30
            print(f'   {k:16s} rel err {np.abs(g - b).max() / sc:.2e}  ref noise {np.abs(a - b).max() / sc:.2e}')
