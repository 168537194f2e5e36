"""Shared test helpers: build product modules for a golden case, run the oracle,
compare against stored arrays / digests."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from cases import MODULE_CASES, NEEDS_A, COT_SEED, tag_seed          # noqa: E402
from params import fill_state_, make_input, make_labels, digest, sample      # noqa: E402

from tam_gcn_amd.graph import ucla, ntu_rgb_d, synthetic             # noqa: E402
from tam_gcn_amd.models import ctrgcn as M                           # noqa: E402
from oracle import ctrgcn_oracle as O                                # noqa: E402

A_BY_V = {20: ucla.Graph().A, 25: ntu_rgb_d.Graph().A, 64: synthetic.Graph().A}


def build_module(kind, kw, V):
    cls = getattr(M, kind)
    if kind in NEEDS_A:
        kw = dict(kw)
        cin, cout = kw.pop('in_channels'), kw.pop('out_channels')
        return cls(cin, cout, A_BY_V[V], **kw)
    return cls(**kw)


def ctrgc_extras(V, device='cpu'):
    A = torch.from_numpy(A_BY_V[V][1].astype(np.float32)) + 0.05 * make_input((V, V), 5)
    return A.to(device).requires_grad_(True), torch.tensor([0.6], device=device, requires_grad=True)


def oracle_run(kind, kw, sd, x, training, extras=None):
    """sd keys are prefixed with 'm.'; returns y."""
    if kind == 'CTRGC':
        A, alpha = extras
        return O.ctrgc(x, sd, 'm', A, alpha)
    if kind == 'unit_gcn':
        if not kw.get('residual', True):
            return O.unit_gcn_noresidual(x, sd, 'm', training)
        return O.unit_gcn(x, sd, 'm', training)
    if kind == 'TemporalConv':
        return O.temporal_conv(x, sd, 'm', kw['kernel_size'], kw.get('stride', 1), kw.get('dilation', 1), training)
    if kind == 'unit_tcn':
        return O.unit_tcn(x, sd, 'm', kw.get('kernel_size', 9), kw.get('stride', 1), training)
    if kind == 'MultiScale_TemporalConv':
        cin, cout, s = kw['in_channels'], kw['out_channels'], kw.get('stride', 1)
        res = kw.get('residual', True)
        mode = 'zero' if not res else ('identity' if cin == cout and s == 1 else 'conv')
        return O.ms_tcn(x, sd, 'm', kw.get('kernel_size', 3), s, tuple(kw.get('dilations', [1, 2, 3, 4])), training,
                        mode, kw.get('residual_kernel_size', 1))
    if kind == 'TCN_GCN_unit':
        return O.tcn_gcn_unit(x, sd, 'm', kw.get('stride', 1), kw.get('residual', True), training)
    raise KeyError(kind)


def golden_get(gold, key):
    """(array, is_digest)"""
    if key in gold.files:
        return gold[key], False
    return gold[key + '#digest'], True


def golden_sample(gold, key, got):
    """(values of `got` at the fixture's 4096 seeded flat indices, the reference's values there) for a digest-stored tensor."""
    return sample(got, key).astype(np.float64), gold[key + '#sample'].astype(np.float64)


def assert_close(name, got, gold, key, rtol, atol):
    ref, is_dig = golden_get(gold, key)
    if is_dig:
        a, b = golden_sample(gold, key, got)                 # element-wise pin: 4096 seeded positions of the interior
        np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=name + ' (sampled elements)')
        g = digest(got)
        scale = max(1.0, float(np.abs(ref[1])) / max(1, got.numel()) ** 0.5)
        # sums: compare with a tolerance scaled by sum|x|; head/tail elementwise
        assert abs(g[0] - ref[0]) <= rtol * abs(ref[1]) + atol * got.numel() ** 0.5, f'{name}: sum {g[0]} vs {ref[0]}'
        assert abs(g[1] - ref[1]) <= rtol * abs(ref[1]) + atol * got.numel() ** 0.5, f'{name}: abs-sum {g[1]} vs {ref[1]}'
        np.testing.assert_allclose(g[3:], ref[3:], rtol=rtol * 10, atol=atol * 10, err_msg=name)
        del scale
    else:
        a = got.detach().cpu().numpy()
        assert a.shape == ref.shape, f'{name}: shape {a.shape} vs {ref.shape}'
        np.testing.assert_allclose(a, ref, rtol=rtol, atol=atol, err_msg=name)
