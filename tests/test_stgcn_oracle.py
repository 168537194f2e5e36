"""CPU: oracle/stgcn_oracle.py against vectors generated from the reference's models/stgcn.py (tests/golden/stgcn.npz), and
the product mirror's constructor / state-dict surface."""
import os

import numpy as np
import pytest
import torch

from cases import STGCN_BLOCK_CASES, STGCN_MODEL_CASES, COT_SEED, tag_seed
from params import fill_state_, make_input, make_labels, digest, sample
from oracle import stgcn_oracle as SO
from tam_gcn_amd.graph import ucla
from tam_gcn_amd.models import stgcn as M

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'stgcn.npz'))
A = torch.tensor(ucla.Graph().A, dtype=torch.float32)


def rmode(kw):
    if not kw.get('residual', True):
        return 'zero'
    return 'identity' if kw['in_channels'] == kw['out_channels'] and kw.get('stride', 1) == 1 else 'conv'


def fill_stgcn_(sd, seed):
    Ab = sd['A'].clone()
    fill_state_(sd, seed)
    r = np.random.RandomState(seed + 17)
    with torch.no_grad():
        for k in sorted(sd.keys()):
            if k.startswith('edge_importance'):
                sd[k].copy_(torch.from_numpy((1 + 0.1 * r.standard_normal(tuple(sd[k].shape))).astype(np.float32)))
        sd['A'].copy_(Ab)


def get(key):
    return (GOLD[key], False) if key in GOLD.files else (GOLD[key + '#digest'], True)


def close(name, got, key, rtol=2e-4, atol=2e-5):
    ref, dig = get(key)
    if dig:
        g = digest(got)
        assert abs(g[1] - ref[1]) <= rtol * abs(ref[1]) + atol * got.numel() ** 0.5, f'{name}: {g[1]} vs {ref[1]}'
        np.testing.assert_allclose(g[3:], ref[3:], rtol=rtol * 10, atol=atol * 10, err_msg=name)
        np.testing.assert_allclose(sample(got, key), GOLD[key + '#sample'], rtol=rtol, atol=atol, err_msg=name + ' (sampled elements)')
    else:
        np.testing.assert_allclose(got.detach().numpy(), ref, rtol=rtol, atol=atol, err_msg=name)


@pytest.mark.parametrize('case', STGCN_BLOCK_CASES, ids=[c[0] for c in STGCN_BLOCK_CASES])
def test_block_oracle(case):
    tag, kw, shape, xseed = case
    blk = M.st_gcn(kw['in_channels'], kw['out_channels'], (9, 3), kw.get('stride', 1), residual=kw.get('residual', True))
    fill_state_(blk.state_dict(), seed=tag_seed(tag))
    sd = {'m.' + k: v.detach().clone().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in blk.state_dict().items()}
    imp = (1 + 0.1 * make_input((3, 20, 20), seed=31)).requires_grad_(True)
    x = make_input(shape, xseed).requires_grad_(True)
    y = SO.st_gcn(x, sd, 'm', A * imp, kw.get('stride', 1), rmode(kw), True)
    (y * make_input(tuple(y.shape), seed=COT_SEED)).sum().backward()
    close('y', y, f'{tag}/y'); close('dx', x.grad, f'{tag}/dx', 1e-3, 1e-4); close('dimp', imp.grad, f'{tag}/dimp', 1e-3, 1e-4)
    for k, _ in blk.named_parameters():
        close(k, sd['m.' + k].grad, f'{tag}/grad/{k}', 2e-3, 2e-4)


@pytest.mark.parametrize('case', STGCN_MODEL_CASES, ids=[c[0] for c in STGCN_MODEL_CASES])
def test_model_oracle_and_surface(case):
    tag, margs, shape = case
    m = M.Model(**margs)
    assert list(m.state_dict().keys()) == [str(k) for k in GOLD[f'{tag}/keys']]          # checkpoint ABI
    fill_stgcn_(m.state_dict(), seed=77)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and 'running' not in k and k != 'A') for k, v in m.state_dict().items()}
    x = make_input(shape, seed=21).requires_grad_(True)
    lab = make_labels(shape[0], margs['num_class'], seed=22)
    logits = SO.model_forward(x, sd, margs['num_point'], True)
    loss = torch.nn.functional.cross_entropy(logits, lab)
    loss.backward()
    np.testing.assert_allclose(logits.detach().numpy(), GOLD[f'{tag}/logits_train'], rtol=1e-3, atol=1e-4)
    assert abs(float(loss) - float(GOLD[f'{tag}/loss'])) <= 1e-4
    np.testing.assert_allclose(x.grad.numpy(), GOLD[f'{tag}/dx'], rtol=5e-3, atol=1e-5 * np.abs(GOLD[f'{tag}/dx']).max() + 1e-7)
    assert [k for k, _ in m.named_parameters()] == [str(k) for k in GOLD[f'{tag}/param_keys']]
    with torch.no_grad():
        le = SO.model_forward(x.detach(), sd, margs['num_point'], False)
        o, f = SO.model_extract_feature(x.detach(), sd, margs['num_point'], False)
    np.testing.assert_allclose(le.numpy(), GOLD[f'{tag}/logits_eval'], rtol=1e-3, atol=1e-3)
    assert list(f.shape) == list(GOLD[f'{tag}/feat_shape']) and list(o.shape) == list(GOLD[f'{tag}/out_shape'])
    assert abs(digest(f)[1] - GOLD[f'{tag}/feat_digest'][1]) <= 1e-3 * abs(GOLD[f'{tag}/feat_digest'][1])


def test_init_rng_order_matches_reference_layout():
    """Same constructor order => torch.manual_seed(s); Model(...) consumes the RNG as the reference does: the mirror builds the same
    modules in the same order with default inits (the reference applies no custom init in stgcn.Model)."""
    torch.manual_seed(1234)
    a = M.Model(**STGCN_MODEL_CASES[0][1]).state_dict()
    got = np.stack([digest(v) for v in a.values()])
    assert np.array_equal(got, GOLD['init_digest'])          # bit-identical initial state-dict
    with pytest.raises(ValueError):
        M.Model(graph=None)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        M.Model(**STGCN_MODEL_CASES[0][1])(torch.zeros(1, 3, 8, 20, 1))
