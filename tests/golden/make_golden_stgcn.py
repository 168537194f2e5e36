"""Golden vectors for the ST-GCN block and model.  Runs ONLY in the build container: imports the
reference's models/stgcn.py + graph/ucla.py and writes data only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_stgcn.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from params import fill_state_, make_input, make_labels, digest, sample  # noqa: E402
from cases import COT_SEED, tag_seed, STGCN_BLOCK_CASES, STGCN_MODEL_CASES  # noqa: E402

sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference')
import graph.ucla                            # noqa: E402,F401  (reference's)
from models import stgcn as R                # noqa: E402      (reference's)

torch.set_num_threads(8)


def fill_stgcn_(sd, seed):
    """fill_state_ plus ST-GCN specifics: edge_importance ~ 1 + 0.1 N(0,1) (fill_state_ would draw N(0,1)); the buffer A is kept."""
    A = sd['A'].clone() if 'A' in sd else None
    fill_state_(sd, seed)
    r = np.random.RandomState(seed + 17)
    with torch.no_grad():
        for k in sorted(sd.keys()):
            if k.startswith('edge_importance'):
                sd[k].copy_(torch.from_numpy((1 + 0.1 * r.standard_normal(tuple(sd[k].shape))).astype(np.float32)))
        if A is not None:
            sd['A'].copy_(A)


def put(out, key, t, full_max=20000):
    if t.numel() <= full_max:
        out[key] = t.detach().cpu().numpy().copy()
    else:
        out[key + '#digest'] = digest(t)
        out[key + '#sample'] = sample(t, key)


def main():
    out = {}
    A = torch.tensor(graph.ucla.Graph().A, dtype=torch.float32)
    for tag, kw, shape, xseed in STGCN_BLOCK_CASES:
        blk = R.st_gcn(kw['in_channels'], kw['out_channels'], (9, 3), kw.get('stride', 1), residual=kw.get('residual', True))
        fill_state_(blk.state_dict(), seed=tag_seed(tag))
        imp = (1 + 0.1 * make_input((3, 20, 20), seed=31)).requires_grad_(True)
        x = make_input(shape, xseed).requires_grad_(True)
        blk.train()
        y, _ = blk(x, A * imp)
        cot = make_input(tuple(y.shape), seed=COT_SEED)
        (y * cot).sum().backward()
        put(out, f'{tag}/y', y); put(out, f'{tag}/dx', x.grad); put(out, f'{tag}/dimp', imp.grad)
        for k, p in blk.named_parameters():
            put(out, f'{tag}/grad/{k}', p.grad)
        for k, b in blk.named_buffers():
            put(out, f'{tag}/buf_after/{k}', b)
        blk.eval()
        with torch.no_grad():
            put(out, f'{tag}/y_eval', blk(x, A * imp)[0])
    torch.manual_seed(1234)
    m0 = R.Model(**STGCN_MODEL_CASES[0][1])
    out['init_digest'] = np.stack([digest(v) for v in m0.state_dict().values()])
    for tag, margs, shape in STGCN_MODEL_CASES:
        m = R.Model(**margs)
        out[f'{tag}/keys'] = np.array(list(m.state_dict().keys()))
        fill_stgcn_(m.state_dict(), seed=77)
        x = make_input(shape, seed=21).requires_grad_(True)
        lab = make_labels(shape[0], margs['num_class'], seed=22)
        m.train()
        logits = m(x)
        loss = torch.nn.functional.cross_entropy(logits, lab)
        loss.backward()
        out[f'{tag}/logits_train'] = logits.detach().numpy()
        out[f'{tag}/loss'] = loss.detach().numpy()
        out[f'{tag}/dx'] = x.grad.numpy()
        out[f'{tag}/param_keys'] = np.array([k for k, _ in m.named_parameters()])
        out[f'{tag}/grad_digest'] = np.stack([digest(p.grad) for _, p in m.named_parameters()])
        for k, p in m.named_parameters():
            if k.startswith('edge_importance') or k.startswith('fcn.'):
                out[f'{tag}/grad/{k}'] = p.grad.numpy()
        m.eval()
        with torch.no_grad():
            out[f'{tag}/logits_eval'] = m(x).numpy()
            o, f = m.extract_feature(x)
            out[f'{tag}/feat_digest'] = digest(f); out[f'{tag}/feat_shape'] = np.array(f.shape)
            out[f'{tag}/out_digest'] = digest(o); out[f'{tag}/out_shape'] = np.array(o.shape)
        print(tag, 'loss', float(loss.detach()))
    np.savez_compressed(os.path.join(HERE, 'stgcn.npz'), **out)
    print('stgcn.npz', len(out), 'arrays')


if __name__ == '__main__':
    main()
