"""-m gpu: the small-batch eval-mode kernel family (csrc/f2.hip, tam_gcn_amd/f2.py) -- what the
reference's inference callers run: ensemble/ensemble_ctrgcn_resnet_eval.py:147-183, models/resnet_gcn_attention.py:82-85,
visual.py:53-55 (model(data) in eval mode on a handful of clips).

Bars: every block, fed the fp64 oracle's own input for that block (teacher-forced), within 2e-5 of max|ref| (exact fp32
MFMA, BatchNorm folded in fp32); logits within 1e-3 of the reference's golden eval logits (test_gpu_model.py's eval section
runs through this path too, the batches there are small) and within 2e-5 of the general eval path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cases import MODEL_CASES, MODEL_PARAM_SEED, MODEL_X_SEED                      # noqa: E402
from params import fill_state_, make_input                                        # noqa: E402
from tam_gcn_amd import f2, _lib                                                    # noqa: E402
from tam_gcn_amd.models import ctrgcn as M                                          # noqa: E402
from oracle import ctrgcn_oracle as O                                               # noqa: E402

DEV = 'cuda:0'


def _model(tag='ucla_t64', seed=MODEL_PARAM_SEED, gold=None, **over):
    """gold: the fixture file -- its running statistics (the reference's, after its training steps on this state) replace
    the seeded ones, which are not the statistics of anything: with those the eval-mode activations grow tenfold per block
    (2e10 at l10) and l6-l8 lose four digits to cancellation on EVERY path (tools/f2_report.py)."""
    margs = dict(next(c for c in MODEL_CASES if c[0] == tag)[1], **over)
    m = M.Model(**margs)
    sd = m.state_dict()
    fill_state_(sd, seed=seed)
    if gold is not None:
        with torch.no_grad():
            for k in sd:
                key = f'{tag}/evalbuf/{k}'
                if 'running_' in k and key in gold.files and tuple(gold[key].shape) == tuple(sd[k].shape):
                    sd[k].copy_(torch.from_numpy(gold[key]))
    return m, margs


@pytest.mark.parametrize('shape', [(2, 3, 52, 20, 1), (1, 3, 13, 20, 1), (2, 3, 30, 20, 2)], ids=['t52', 't13_ragged', 't30_two_persons'])
def test_every_block_against_the_fp64_oracle(shape, golden_models):
    """T = 52 is the reference's clip length (tiles of 4 frames: 13 / 7 (ragged) / 4 per depth); T = 13 leaves 1-, 3- and
    4-frame tiles; T = 30 with two persons: 30 -> 15 -> 8 frames, 4 clip-persons."""
    m, margs = _model('ucla_t52', gold=golden_models, num_person=shape[4])
    sd64 = {k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in m.state_dict().items()}
    x = make_input(shape, seed=MODEL_X_SEED)
    h, N, Mp = O._stem(x.double(), sd64, 20, False)
    ins, outs = [], []
    for i in range(1, 11):
        ins.append(h)
        h = O.tcn_gcn_unit(h, sd64, f'l{i}', O._STRIDES.get(i, 1), residual=(i != 1), training=False)
        outs.append(h)
    m = m.to(DEV).eval()
    eng = f2.FusedEval(m)
    blocks = eng._packed(torch.device(DEV))
    for i, (b, xin, ref) in enumerate(zip(blocks, ins, outs), 1):
        got = eng._block(b, xin.float().to(DEV).contiguous()).double().cpu()
        assert got.shape == ref.shape, (i, got.shape, ref.shape)
        err = float((got - ref).abs().max() / ref.abs().max())
        assert err <= 2e-5, f'l{i}: {err:.3e} of max|ref|'
    with torch.no_grad():
        logits = eng(x.to(DEV)).double().cpu()
    ref = O.model_forward(x.double(), sd64, 20, training=False)
    assert float((logits - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    assert torch.equal(logits.argmax(1), ref.argmax(1))


def test_model_forward_routes_small_eval_batches_here(monkeypatch, golden_models):
    """Model.forward / extract_feature take this path in eval mode without autograd for <= F2_MAX_CLIPS clip-persons, and the
    general eval path otherwise (larger batches, TAMGCN_F2=0, grad mode, train mode); both agree to fp32 rounding and
    with the reference's golden eval logits."""
    tag, margs, shape = next(c for c in MODEL_CASES if c[0] == 'ucla_t52')
    m, _ = _model(tag, gold=golden_models)
    m = m.to(DEV).eval()
    x = make_input(shape, seed=MODEL_X_SEED).to(DEV)
    calls = []
    real = f2.FusedEval.blocks
    monkeypatch.setattr(f2.FusedEval, 'blocks', lambda self, x: (calls.append(1), real(self, x))[1])
    with torch.no_grad():
        a = m(x)
        fa, _ = m.extract_feature(x)
    assert len(calls) == 2
    assert np.abs(a.cpu().numpy() - golden_models[f'{tag}/logits_eval']).max() <= 1e-3
    monkeypatch.setenv('TAMGCN_F2', '0')
    with torch.no_grad():
        b = m(x)
        fb, _ = m.extract_feature(x)
    assert len(calls) == 2
    assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
    assert float((fa - fb).abs().max()) <= 2e-5 * float(fb.abs().max())
    monkeypatch.setenv('TAMGCN_F2', '1')
    big = make_input((f2.F2_MAX_CLIPS + 1,) + tuple(shape[1:]), seed=3).to(DEV)
    with torch.no_grad():
        m(big)                                              # too many clips: general path
    assert len(calls) == 2
    m(x)                                                    # grad mode: general path (autograd)
    assert len(calls) == 2
    m.train()
    with torch.no_grad():
        m(x)
    assert len(calls) == 2
    m25 = M.Model(**next(c for c in MODEL_CASES if c[0] == 'ntu_t20')[1]).to(DEV).eval()
    with torch.no_grad():
        assert m25._f2(torch.zeros(1, 3, 16, 25, 2, device=DEV)) is None      # V = 25: outside the family, served by the general path
        assert m25(torch.randn(1, 3, 16, 25, 2, device=DEV)).shape[0] == 1


def test_refolds_after_every_kind_of_state_change(golden_models):
    """The folded weights follow in-place parameter updates, load_state_dict and the running statistics a train-mode forward
    rewrites through raw pointers."""
    m, _ = _model('ucla_t52', gold=golden_models)
    m = m.to(DEV).eval()
    x = make_input((2, 3, 52, 20, 1), seed=MODEL_X_SEED).to(DEV)

    def pair():
        import os
        with torch.no_grad():
            a = m(x)
            os.environ['TAMGCN_F2'] = '0'
            try:
                b = m(x)
            finally:
                os.environ['TAMGCN_F2'] = '1'
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
        return a
    a0 = pair()
    with torch.no_grad():
        m.l3.tcn1.branches[0][1].weight.mul_(1.5)
        m.l6.gcn1.convs[1].conv4.bias.add_(0.3)
    a1 = pair()
    assert float((a1 - a0).abs().max()) > 0
    m.train()
    m(make_input((4, 3, 52, 20, 1), seed=5).to(DEV) * 3 + 0.5).sum().backward()
    m.eval()
    a2 = pair()
    assert float((a2 - a1).abs().max()) > 1e-3 * float(a1.abs().max())
    m2, _ = _model('ucla_t52', gold=golden_models)
    with torch.no_grad():
        for p in m2.parameters():
            p.mul_(0.9)
    m.load_state_dict(m2.state_dict())
    a3 = pair()
    assert float((a3 - a2).abs().max()) > 1e-3 * float(a2.abs().max())


def test_refolds_after_a_flat_arena_step(golden_models):
    """with a ParamArena the optimiser writes the flat buffer (no p._version bump); an eval batch of <= 32 clips
    after such a step must use the NEW weights: f2 == the general eval path, and both differ from before the step."""
    from tam_gcn_amd.distributed import ParamArena, SGDNesterov
    m, _ = _model('ucla_t52', gold=golden_models)
    m = m.to(DEV).eval()
    arena = ParamArena(m)
    bucket = arena.grad_bucket()
    opt = SGDNesterov(arena.params, lr=0.05, momentum=0.9, weight_decay=1e-4, arena=arena, bucket=bucket)
    x = make_input((2, 3, 52, 20, 1), seed=MODEL_X_SEED).to(DEV)

    def pair():
        import os
        with torch.no_grad():
            a = m(x)
            os.environ['TAMGCN_F2'] = '0'
            try:
                b = m(x)
            finally:
                os.environ['TAMGCN_F2'] = '1'
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
        return a
    a0 = pair()
    g = torch.Generator().manual_seed(3)
    for _ in range(2):                                     # eval mode throughout: no train-mode BatchNorm forward in between
        for p in arena.params:
            p.grad = (torch.randn(p.shape, generator=g) * p.detach().abs().mean().cpu()).to(DEV)
        bucket.pack()
        opt.step()
        a1 = pair()
        assert float((a1 - a0).abs().max()) > 1e-3 * float(a0.abs().max())
        a0 = a1


def test_graph_replay_and_launch_count():
    """inference.GraphedForward captures this path: replay = eager bit for bit; 53 ABI launches per forward (5 per block,
    stem, pool + fc) against 118 of the general eval path."""
    from tam_gcn_amd.inference import GraphedForward
    m, _ = _model()
    m = m.to(DEV).eval()
    fast = GraphedForward(m)
    for nb in (1, 4):
        x = make_input((nb, 3, 52, 20, 1), seed=nb).to(DEV)
        with torch.no_grad():
            ref = m(x)
        assert torch.equal(fast(x).clone(), ref)

    class Count:
        def __init__(self, lib):
            self.lib, self.n, self.names = lib, 0, []

        def __getattr__(self, name):
            fn = getattr(self.lib, name)
            if not name.startswith('tamgcn_') or name in ('tamgcn_last_error',):
                return fn

            def w(*args):
                self.n += 1
                self.names.append(name)
                return fn(*args)
            return w
    real = _lib.load()
    cnt = Count(real)
    _lib._lib = cnt
    try:
        with torch.no_grad():
            m(make_input((1, 3, 52, 20, 1), seed=9).to(DEV))
    finally:
        _lib._lib = real
    assert cnt.n <= 56, (cnt.n, cnt.names)
    assert sum(n.startswith('tamgcn_f2_') for n in cnt.names) == 50


def test_block_is_a_registered_operator():
    """torch.ops.tamgcn.tcn_gcn_unit_eval: schema and fake-tensor checks (torch.library.opcheck), traced by torch.compile
    without a graph break."""
    m, _ = _model()
    m = m.to(DEV).eval()
    eng = f2.FusedEval(m)
    blocks = eng._packed(torch.device(DEV))
    x = make_input((2, 64, 12, 20), seed=3).to(DEV)
    for b in (blocks[1], blocks[4]):                        # identity residual; 64 -> 128, stride 2, convolutional residuals
        torch.library.opcheck(torch.ops.tamgcn.tcn_gcn_unit_eval.default, (x, None, b.params, b.geom),
                              test_utils=('test_schema', 'test_faketensor'))
    b = blocks[4]

    def f(x):
        out, xp = torch.ops.tamgcn.tcn_gcn_unit_eval(x, None, b.params, b.geom)
        return out * 2, xp
    with torch.no_grad():
        ref = f(x)
        got = torch.compile(f, fullgraph=True, backend='eager')(x)
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1]) and tuple(ref[0].shape) == (2, 128, 6, 20)
    assert float((ref[1].sum(1) - ref[0].sum(2) / 2).abs().max()) <= 1e-5 * float(ref[0].abs().max()) * 6     # xpart = frame sums of out


def test_graphed_forward_batch_slices_on_streams():
    """GraphedForward(split=2): the eval batch in two slices on two streams inside one graph equals the whole-batch forward
    (same kernels on 64-clip slices: to rounding of nothing -- eval BatchNorm is per element) and replays identically."""
    from tam_gcn_amd.inference import GraphedForward
    m, _ = _model()
    m = m.to(DEV).eval()
    x = make_input((128, 3, 32, 20, 1), seed=2).to(DEV)
    with torch.no_grad():
        ref = m(x)
    fast = GraphedForward(m, split=2)
    a = fast(x).clone()
    b = fast(x).clone()
    assert torch.equal(a, b)
    assert float((a - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert GraphedForward(m, split=4)(x[:100]).shape == ref[:100].shape       # 100 clips: one slice of >= 64, i.e. unsplit


def test_engine_argument_guards():
    m, _ = _model()
    with pytest.raises(ValueError):
        f2.FusedEval(m.to(DEV))                             # train mode
    m.eval()
    eng = f2.FusedEval(m)
    x = make_input((1, 3, 52, 20, 1), seed=1)
    with torch.no_grad():
        with pytest.raises(RuntimeError, match='no CPU path'):
            eng(x)
    with pytest.raises(RuntimeError, match='no_grad'):
        eng(x.to(DEV))
    lib = _lib.load()
    d = _lib.F2GemmDesc(N=1, K=64, M=60, T=8, V=20, mode=1, relu_rows=0, x=1 << 20, w=1 << 20, b=1 << 20, add=None, out=1 << 20)
    import ctypes as C
    assert lib.tamgcn_f2_gemm(C.byref(d), None) != 0 and b'M %' in lib.tamgcn_last_error()
    d = _lib.F2GemmDesc(N=1, K=64, M=64, T=8, V=25, mode=1, relu_rows=0, x=1 << 20, w=1 << 20, b=1 << 20, add=None, out=1 << 20)
    assert lib.tamgcn_f2_gemm(C.byref(d), None) != 0 and b'V = 20' in lib.tamgcn_last_error()
