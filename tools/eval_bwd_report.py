"""GPU-box diagnostic: gradients through the model in eval() mode against the fp64 oracle (the arrangement of
tests/test_gpu_model.py::test_backward_through_eval_mode_matches_the_oracle), per tensor, worst first; also in train mode."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from cases import MODEL_CASES, MODEL_PARAM_SEED, MODEL_X_SEED                  # noqa: E402
from params import fill_state_, make_input                                    # noqa: E402
from tam_gcn_amd.models import ctrgcn as M                                     # noqa: E402
from oracle import ctrgcn_oracle as O                                          # noqa: E402

gold = np.load(os.path.join(ROOT, 'tests', 'golden', 'models.npz'))
tagsel = sys.argv[1] if len(sys.argv) > 1 else 'ucla_t13'
tag, margs, shape = next(c for c in MODEL_CASES if c[0] == tagsel)
dev = torch.device('cuda:0')
for training in (False, True):
    m = M.Model(**margs)
    fill_state_(m.state_dict(), seed=MODEL_PARAM_SEED)
    sd0 = m.state_dict()
    with torch.no_grad():
        for k in sd0:
            if 'running_' in k:
                sd0[k].copy_(torch.from_numpy(gold[f'{tag}/evalbuf/{k}']))
    sd = {k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in m.state_dict().items()}
    for k, _ in m.named_parameters():
        sd[k].requires_grad_(True)
    xo = make_input(shape, seed=MODEL_X_SEED).double().requires_grad_(True)
    cot = make_input((shape[0], margs['num_class']), seed=77).double()
    lo = O.model_forward(xo, sd, margs['num_point'], training=training)
    (lo * cot).sum().backward()
    m = m.to(dev)
    m.train(training)
    x = make_input(shape, seed=MODEL_X_SEED).to(dev).requires_grad_(True)
    lg = m(x)
    (lg * cot.float().to(dev)).sum().backward()
    torch.cuda.synchronize()
    rows = []

    def rep(name, a, b):
        a, b = a.detach().cpu().double(), b.detach().double()
        sc = float(b.abs().max()) + 1e-30
        err = (a - b).abs()
        rows.append((float(err.max()) / sc, float((err > 2e-4 * sc).double().mean()), float((a - b).norm() / (b.norm() + 1e-30)), name, a.numel()))
    rep('logits', lg, lo)
    rep('dx', x.grad, xo.grad)
    for k, p in m.named_parameters():
        if float(sd[k].grad.abs().max()) > 1e-12:
            rep(k, p.grad, sd[k].grad)
    print(f'--- {tag} training={training}: max rel err | share of entries beyond 2e-4 | rel L2 | tensor (elements)')
    for r in rows[:2] + sorted(rows[2:], reverse=True)[:14]:
        print(f'   {r[0]:.2e} | {r[1]:7.2%} | {r[2]:.2e} | {r[3]} ({r[4]})')
    per_layer = {}
    for r in rows[2:]:
        per_layer.setdefault(r[3].split('.')[0], []).append(r[2])
    print('   rel L2 per layer (max over its tensors): ' + ', '.join(f'{k}: {max(v):.1e}' for k, v in per_layer.items()))
