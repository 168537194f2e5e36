"""GPU-box diagnostic: ad-hoc module cases (not in tests/golden) on the HIP path against the CPU oracle -- forward,
input gradient and every parameter gradient, max-abs error relative to max|ref|.

    python tools/module_case_report.py        # the stride-2 V = 64 pieces of TCN_GCN_unit(64, 128, stride 2) at T = 70
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from helpers import build_module, oracle_run, fill_state_, make_input, COT_SEED     # noqa: E402

dev = torch.device('cuda:0')
CASES = [
    ('unit_tcn k1 s2 64->128 V64 T70', 'unit_tcn', dict(in_channels=64, out_channels=128, kernel_size=1, stride=2), (1, 64, 70, 64)),
    ('tconv k5 s2 d1 32 V64 T70', 'TemporalConv', dict(in_channels=32, out_channels=32, kernel_size=5, stride=2, dilation=1), (1, 32, 70, 64)),
    ('tconv k5 s2 d2 32 V64 T70', 'TemporalConv', dict(in_channels=32, out_channels=32, kernel_size=5, stride=2, dilation=2), (1, 32, 70, 64)),
    ('tconv k5 s2 d2 32 V64 T16', 'TemporalConv', dict(in_channels=32, out_channels=32, kernel_size=5, stride=2, dilation=2), (1, 32, 16, 64)),
    ('mstcn 128 s2 V64 T70', 'MultiScale_TemporalConv', dict(in_channels=128, out_channels=128, kernel_size=5, stride=2, dilations=[1, 2], residual=False), (1, 128, 70, 64)),
    ('mstcn 128 s2 V64 T16', 'MultiScale_TemporalConv', dict(in_channels=128, out_channels=128, kernel_size=5, stride=2, dilations=[1, 2], residual=False), (1, 128, 16, 64)),
    ('mstcn 128 s1 V64 T70', 'MultiScale_TemporalConv', dict(in_channels=128, out_channels=128, kernel_size=5, stride=1, dilations=[1, 2], residual=False), (1, 128, 70, 64)),
    ('mstcn 128 s2 V20 T70', 'MultiScale_TemporalConv', dict(in_channels=128, out_channels=128, kernel_size=5, stride=2, dilations=[1, 2], residual=False), (1, 128, 70, 20)),
    ('gcn 64->128 V64 T70', 'unit_gcn', dict(in_channels=64, out_channels=128), (1, 64, 70, 64)),
    ('unit 64->128 s2 V64 T16', 'TCN_GCN_unit', dict(in_channels=64, out_channels=128, stride=2, residual=True), (1, 64, 16, 64)),
    ('unit 64->128 s2 V64 T32', 'TCN_GCN_unit', dict(in_channels=64, out_channels=128, stride=2, residual=True), (1, 64, 32, 64)),
    ('unit 64->128 s2 V64 T34', 'TCN_GCN_unit', dict(in_channels=64, out_channels=128, stride=2, residual=True), (1, 64, 34, 64)),
    ('unit 64->128 s2 V64 T70', 'TCN_GCN_unit', dict(in_channels=64, out_channels=128, stride=2, residual=True), (1, 64, 70, 64)),
    ('unit 64->128 s1 V64 T70', 'TCN_GCN_unit', dict(in_channels=64, out_channels=128, stride=1, residual=True), (1, 64, 70, 64)),
    ('unit 64->128 s2 V64 T70 nores', 'TCN_GCN_unit', dict(in_channels=64, out_channels=128, stride=2, residual=False), (1, 64, 70, 64)),
    ('unit 64->128 s2 V20 T70', 'TCN_GCN_unit', dict(in_channels=64, out_channels=128, stride=2, residual=True), (1, 64, 70, 20)),
    ('unit 64->64 s2 V64 T70', 'TCN_GCN_unit', dict(in_channels=64, out_channels=64, stride=2, residual=True), (1, 64, 70, 64)),
]
if len(sys.argv) > 1:
    CASES = [c for c in CASES if sys.argv[1] in c[0]]
for name, kind, kw, shape in CASES:
    mod = build_module(kind, kw, shape[-1])
    fill_state_(mod.state_dict(), seed=1234)
    sd = {'m.' + k: v.detach().clone() for k, v in mod.state_dict().items()}
    for k, _ in mod.named_parameters():
        sd['m.' + k].requires_grad_(True)
    xo = make_input(shape, 77).requires_grad_(True)
    yo = oracle_run(kind, kw, sd, xo, True)
    cot = make_input(tuple(yo.shape), COT_SEED)
    (yo * cot).sum().backward()
    mod = mod.to(dev).train()
    x = make_input(shape, 77).to(dev).requires_grad_(True)
    try:
        y = mod(x)
        (y * cot.to(dev)).sum().backward()
        torch.cuda.synchronize()
    except Exception as e:                                   # noqa: BLE001
        print(f'{name}: FAILED {type(e).__name__}: {e}')
        continue
    rel = lambda a, b: float((a.detach().cpu().double() - b.detach().double()).abs().max() / (b.detach().double().abs().max() + 1e-30))   # noqa: E731
    rows = [('y', rel(y, yo)), ('dx', rel(x.grad, xo.grad))]
    for k, p in mod.named_parameters():
        go = sd['m.' + k].grad
        if float(go.abs().max()) > 1e-3:
            rows.append((k, rel(p.grad, go)))
    bad = [(k, e) for k, e in rows if e > 1e-3]
    print(f'{name}: y {rows[0][1]:.1e} dx {rows[1][1]:.1e} worst grad {max(e for _, e in rows[2:]):.1e}' + (f'   BAD: {bad[:6]}' if bad else ''))
