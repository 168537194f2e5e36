"""Pins the oracle (oracle/ctrgcn_oracle.py) against golden vectors produced by
importing the reference's own models/ctrgcn.py (tests/golden/make_golden.py).
Also pins the product modules' state-dict key set: parameters are regenerated
per key, so any key/shape drift shows up as a numeric mismatch."""
import numpy as np
import pytest
import torch

from helpers import (MODULE_CASES, COT_SEED, tag_seed, fill_state_, make_input, build_module, ctrgc_extras,
                     oracle_run, assert_close, O)

RTOL, ATOL = 2e-4, 2e-5


@pytest.mark.parametrize('case', MODULE_CASES, ids=[c[0] for c in MODULE_CASES])
def test_module_case(case, golden_modules):
    tag, kind, kw, shape, xseed = case
    gold = golden_modules
    mod = build_module(kind, kw, shape[-1])
    fill_state_(mod.state_dict(), seed=tag_seed(tag))
    sd = {'m.' + k: v.detach().clone() for k, v in mod.state_dict().items()}
    pnames = [k for k, _ in mod.named_parameters()]
    for k in pnames:
        sd['m.' + k].requires_grad_(True)
    x = make_input(shape, xseed).requires_grad_(True)
    extras = ctrgc_extras(shape[-1]) if kind == 'CTRGC' else None
    y = oracle_run(kind, kw, sd, x, True, extras)
    cot = make_input(tuple(y.shape), COT_SEED)
    (y * cot).sum().backward()
    assert_close('y', y, gold, f'{tag}/y', RTOL, ATOL)
    assert_close('dx', x.grad, gold, f'{tag}/dx', RTOL * 5, ATOL * 5)
    for k in pnames:
        g = sd['m.' + k].grad
        assert g is not None, k
        scale = float(g.abs().max()) + 1e-6
        assert_close(f'grad {k}', g, gold, f'{tag}/grad/{k}', 2e-3, 2e-4 * max(1.0, scale))
    for k, _ in mod.named_buffers():
        assert_close(f'buf {k}', sd['m.' + k].detach().float(), gold, f'{tag}/buf_after/{k}', 1e-4, 1e-5)
    if kind == 'CTRGC':
        A, alpha = extras
        np.testing.assert_allclose(A.grad.numpy(), gold[f'{tag}/dA'], rtol=2e-3, atol=2e-4)
        np.testing.assert_allclose(alpha.grad.numpy(), gold[f'{tag}/dalpha'], rtol=2e-3, atol=2e-3)
    # eval-mode forward with the post-step running stats
    with torch.no_grad():
        sde = {k: v.detach() for k, v in sd.items()}
        ye = oracle_run(kind, kw, sde, x.detach(), False,
                        tuple(t.detach() for t in extras) if extras else None)
    assert_close('y_eval', ye, gold, f'{tag}/y_eval', RTOL, ATOL)


def test_mean_commuted_restatement_matches(golden_modules):
    """The kernel specification (conv of the T-mean) equals the reference order of
    operations (mean of the conv) to fp32 rounding."""
    tag, kind, kw, shape, xseed = MODULE_CASES[1]
    mod = build_module(kind, kw, shape[-1])
    fill_state_(mod.state_dict(), seed=tag_seed(tag))
    sd = {'m.' + k: v.detach() for k, v in mod.state_dict().items()}
    x = make_input(shape, xseed)
    A, alpha = ctrgc_extras(shape[-1])
    a = O.ctrgc(x, sd, 'm', A.detach(), alpha.detach())
    b = O.ctrgc_mean_commuted(x, sd, 'm', A.detach(), alpha.detach())
    assert float((a - b).abs().max()) < 1e-4 * float(a.abs().max())
