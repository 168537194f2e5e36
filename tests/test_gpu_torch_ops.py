"""-m gpu: the torch.library registration (tam_gcn_amd/torch_ops.py): schema / fake-tensor / autograd-registration checks by
torch.library.opcheck, tracing without a graph break (torch.compile, fullgraph), and equality with the autograd.Function
forms the registered ops replaced."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from params import make_input, make_labels            # noqa: E402
from tam_gcn_amd import functional as Fn, torch_ops    # noqa: E402,F401
from tam_gcn_amd.models import ctrgcn as M             # noqa: E402


def _ctrgc_args(dev, V=20, Cin=64, Cout=64, R=8, N=2, T=8):
    g = torch.Generator().manual_seed(1)
    r = lambda *s: (torch.rand(*s, generator=g) * 2 - 1).to(dev)        # noqa: E731
    return [r(N, Cin, T, V), r(V, V) * 0.2, torch.tensor([0.6], device=dev), r(R, Cin, 1, 1) * 0.2, r(R) * 0.1, r(R, Cin, 1, 1) * 0.2,
            r(R) * 0.1, r(Cout, Cin, 1, 1) * 0.2, r(Cout) * 0.1, r(Cout, R, 1, 1) * 0.3, r(Cout) * 0.1]


def test_opcheck_and_registration():
    dev = torch.device('cuda:0')
    args = _ctrgc_args(dev)
    for t in args:
        t.requires_grad_(t.is_floating_point())
    utils = ('test_schema', 'test_faketensor', 'test_autograd_registration')
    torch.library.opcheck(torch.ops.tamgcn.ctrgc.default, tuple(args), test_utils=utils)
    logits = make_input((7, 10), 3).to(dev).requires_grad_(True)
    torch.library.opcheck(torch.ops.tamgcn.cross_entropy.default, (logits, make_labels(7, 10, 4).to(dev)), test_utils=utils)
    x = make_input((2, 32, 6, 20), 5).to(dev).requires_grad_(True)
    w, b = make_input((4, 32, 1, 1), 6).to(dev).requires_grad_(True), make_input((4,), 7).to(dev).requires_grad_(True)
    torch.library.opcheck(torch.ops.tamgcn.pointwise_conv.default, (x, w, b), test_utils=utils)
    W, bb = make_input((10, 32), 8).to(dev).requires_grad_(True), make_input((10,), 9).to(dev).requires_grad_(True)
    torch.library.opcheck(torch.ops.tamgcn.head.default, (x, W, bb, 1), test_utils=utils)
    parent = torch.arange(20, dtype=torch.int32, device=dev).roll(1)
    torch.library.opcheck(torch.ops.tamgcn.stream_derive.default, (make_input((2, 3, 8, 20, 1), 2).to(dev), parent, 1),
                          test_utils=('test_schema', 'test_faketensor'))


def test_registered_ops_equal_the_function_forms_and_trace_without_graph_break():
    dev = torch.device('cuda:0')
    args = _ctrgc_args(dev)
    a1 = [t.clone().requires_grad_(t.is_floating_point()) for t in args]
    a2 = [t.clone().requires_grad_(t.is_floating_point()) for t in args]
    y1 = torch.ops.tamgcn.ctrgc(*a1)
    y2 = Fn.CTRGCFn.run(*a2)
    cot = make_input(tuple(y1.shape), 11).to(dev)
    (y1 * cot).sum().backward()
    (y2 * cot).sum().backward()
    assert torch.equal(y1, y2)
    for i, (u, v) in enumerate(zip(a1, a2)):      # the x3-recomputing dE kernel accumulates dp / dq with float atomics: equal to rounding
        err = float((u.grad - v.grad).abs().max()) / (float(v.grad.abs().max()) + 1e-30)
        assert err <= 2e-5, f'gradient {i}: {err:.2e}' 

    mod = M.CTRGC(64, 64).to(dev)
    ce = Fn.CrossEntropyLoss()

    def f(x, A, alpha, lab):
        y = mod(x, A, alpha)                                       # torch.ops.tamgcn.ctrgc
        logits = torch.ops.tamgcn.head(y, mod.conv3.weight.view(64, 64)[:10].contiguous(), mod.conv3.bias[:10].contiguous(), 1)
        return ce(logits, lab)                                     # torch.ops.tamgcn.cross_entropy

    x, A, alpha = args[0].clone().requires_grad_(True), args[1].clone(), args[2].clone()
    lab = make_labels(2, 10, 4).to(dev)
    eager = f(x, A, alpha, lab)
    compiled = torch.compile(f, backend='eager', fullgraph=True)(x, A, alpha, lab)     # fullgraph: any graph break raises
    assert torch.allclose(eager, compiled)
    compiled.backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()


def test_head_pooled_equals_head_and_block_row_means():
    """tamgcn::head_pooled with the row means the last block's final pass emits == tamgcn::head pooling x itself
    (models/ctrgcn.py:343-348), forward and every gradient; two bodies per clip and one."""
    from tam_gcn_amd import ops
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(5)
    for M in (2, 1):
        N, Cc, T, V, K = 3, 24, 5, 25, 7
        pre = torch.randn(N * M, Cc, T, V, generator=g).to(dev)
        W = torch.randn(K, Cc, generator=g).to(dev).requires_grad_(True)
        b = torch.randn(K, generator=g).to(dev).requires_grad_(True)
        x, rm = ops.add_act_fwd(ops.S(pre), None, True, Cc, rowmean=True)
        assert torch.allclose(rm, x.mean((2, 3)), rtol=1e-5, atol=1e-6)
        assert torch.equal(x, ops.add_act_fwd(ops.S(pre), None, True, Cc))
        x1 = x.clone().requires_grad_(True)
        x2 = x.clone().requires_grad_(True)
        cot = torch.randn(N, K, generator=g).to(dev)
        y1 = torch.ops.tamgcn.head(x1, W, b, M)
        g1 = torch.autograd.grad((y1 * cot).sum(), (x1, W, b))
        y2 = torch.ops.tamgcn.head_pooled(x2, rm, W, b, M)
        g2 = torch.autograd.grad((y2 * cot).sum(), (x2, W, b))
        assert torch.allclose(y1, y2, rtol=1e-5, atol=1e-6)
        for a, c in zip(g1, g2):
            assert torch.allclose(a, c, rtol=1e-5, atol=1e-6)
