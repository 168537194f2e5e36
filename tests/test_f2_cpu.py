"""CPU: the host side of the small-batch eval family (tam_gcn_amd/f2.py) -- BatchNorm folding and operand packing -- against the
oracle.  The kernels' contracts (include/tamgcn.h: tamgcn_f2_e / _f2_gcn / _f2_gemm / _f2_tcn) are restated here in plain torch
on the FOLDED tensors `_Block` hands them; the result must equal the oracle's eval-mode TCN_GCN_unit (reference
models/ctrgcn.py:266-284) on the unfolded state.  (The kernels themselves: tests/test_gpu_f2.py.)"""
import pytest
import torch
import torch.nn.functional as F

from cases import MODEL_CASES, MODEL_PARAM_SEED
from params import fill_state_, make_input
from tam_gcn_amd import f2
from tam_gcn_amd.models import ctrgcn as M
from oracle import ctrgcn_oracle as O


def _restate(b, x):
    """What the five launches of one block compute, from the block's folded parameter list (f2._Block.params / .geom)."""
    (W12, B12, W3, B3, W4, B4, PA, alpha, sy, ty, Wd, bd, Wo, bo, We, be, sp, tp, Wr, br), rest = b.params[:20], b.params[20:]
    R, gmode, Cb, nb, ks, stride, rmode = b.geom[:7]
    dils = b.geom[7:7 + nb]
    N, Cin, T, V = x.shape
    Cout = W3.shape[0] // 3
    # tamgcn_f2_e
    xbar = x.mean(2)                                                              # (N, Cin, V)
    pq = torch.einsum('kc,ncv->nkv', W12, xbar) + B12[None, :, None]              # rows s*2R + [p | q]
    pq = pq.view(N, 3, 2, R, V)
    D = torch.tanh(pq[:, :, 0, :, :, None] - pq[:, :, 1, :, None, :])             # (N, 3, R, V, V)
    E = alpha * (torch.einsum('scr,nsruv->nscuv', W4, D) + B4[None, :, :, None, None]) + PA[None, :, None]
    # tamgcn_f2_gcn
    x3 = (torch.einsum('kc,nctv->nktv', W3, x) + B3[None, :, None, None]).view(N, 3, Cout, T, V)
    z = torch.einsum('nscuv,nsctv->nctu', E, x3)
    y = sy[None, :, None, None] * z + ty[None, :, None, None]
    res = 0 if gmode == 0 else x if gmode == 1 else torch.einsum('kc,nctv->nktv', Wd, x) + bd[None, :, None, None]
    sm, df = y + res, res - y
    # tamgcn_f2_gemm mode 0, mode 1
    g = torch.relu(sm + torch.tanh(torch.einsum('kc,nctv->nktv', Wo, df) + bo[None, :, None, None]))
    h = torch.einsum('kc,nctv->nktv', We, g) + be[None, :, None, None]
    Ch = (nb + 1) * Cb
    h = torch.cat((torch.relu(h[:, :Ch]), h[:, Ch:]), 1)
    # tamgcn_f2_tcn
    outs = []
    for i in range(nb):
        w = rest[2 * i].view(Cb, Cb, ks, 1)
        pad = (ks - 1) * dils[i] // 2
        outs.append(F.conv2d(h[:, i * Cb:(i + 1) * Cb], w, rest[2 * i + 1], stride=(stride, 1), padding=(pad, 0), dilation=(dils[i], 1)))
    pool = F.max_pool2d(h[:, nb * Cb:Ch], kernel_size=(3, 1), stride=(stride, 1), padding=(1, 0))
    outs.append(sp[None, :, None, None] * pool + tp[None, :, None, None])
    outs.append(h[:, Ch:, ::stride])
    out = torch.cat(outs, 1)
    if rmode == 1:
        out = out + x
    elif rmode == 2:
        out = out + torch.einsum('kc,nctv->nktv', Wr, x[:, :, ::stride]) + br[None, :, None, None]
    return torch.relu(out)


@pytest.mark.parametrize('T', [13, 52])
def test_folded_blocks_equal_the_oracle(T):
    margs = MODEL_CASES[1][1]
    m = M.Model(**margs).double()
    sd = m.state_dict()
    fill_state_(sd, seed=MODEL_PARAM_SEED)
    with torch.no_grad():                                   # moderate running statistics (seeded ones blow the activations up tenfold per block)
        for k, v in sd.items():
            if k.endswith('running_var'):
                v.mul_(4.0)
    m.eval()
    x = make_input((2, 3, T, 20, 1), seed=5).double()
    h, _, _ = O._stem(x, sd, 20, False)
    for i in range(1, 11):
        blk = f2._Block(getattr(m, f'l{i}'), torch.device('cpu'))
        ref = O.tcn_gcn_unit(h, sd, f'l{i}', O._STRIDES.get(i, 1), residual=(i != 1), training=False)
        with torch.no_grad():
            got = _restate(blk, h)
        assert got.shape == ref.shape
        assert float((got - ref).abs().max()) <= 1e-10 * float(ref.abs().max()), f'l{i}'
        h = ref


def test_unsupported_geometries_are_refused_before_any_launch():
    with pytest.raises(f2.Unsupported):
        f2.FusedEval(M.Model(**MODEL_CASES[3][1]).eval())   # NTU: 25 joints
    blk = M.TCN_GCN_unit(65, 65, M.Model(**MODEL_CASES[1][1]).graph.A, kernel_size=5, dilations=[1, 2, 3])
    with pytest.raises(f2.Unsupported):                    # 3 temporal branches of 13 channels: not a multiple of 16
        f2._Block(blk, torch.device('cpu'))


def test_state_key_sees_updates_through_a_param_arena():
    """parameters inside a ParamArena are no_grad views of its flat buffer, so flat SGD / a flat broadcast /
    a load into the buffer leave every p._version untouched.  The re-fold key must change all the same."""
    from tam_gcn_amd.distributed import ParamArena, SGDNesterov
    m = M.Model(**MODEL_CASES[1][1]).eval()
    eng = f2.FusedEval(m)
    k_plain = eng._state_key()
    arena = ParamArena(m)
    versions = [p._version for p in m.parameters()]
    k0 = eng._state_key()
    assert k0 != k_plain                                     # p.data was re-pointed at the arena
    with torch.no_grad():
        arena.flat.mul_(1.01)                                # e.g. a load into the flat buffer
    assert [p._version for p in m.parameters()] == versions  # ... which the parameters' own counters do not see
    k1 = eng._state_key()
    assert k1 != k0
    bucket = arena.grad_bucket()
    for p in arena.params:
        p.grad = torch.ones_like(p)
    bucket.pack()
    SGDNesterov(arena.params, lr=0.1, arena=arena, bucket=bucket).step()
    k2 = eng._state_key()
    assert k2 != k1
    arena.touch()                                            # what a caller does after replaying a captured optimiser step
    assert eng._state_key() != k2
    # the general eval path's BatchNorm-coefficient cache (functional._eval_cached) keys on the arena too
    from tam_gcn_amd import functional as Fn
    bn = Fn.BN(m.l1.tcn1.branches[0][1])
    built = []
    Fn._eval_cached(m.l1.tcn1, 't', [bn], lambda: built.append(1))
    Fn._eval_cached(m.l1.tcn1, 't', [bn], lambda: built.append(1))
    assert len(built) == 1
    with torch.no_grad():
        arena.flat.add_(0.5)
    Fn._eval_cached(m.l1.tcn1, 't', [bn], lambda: built.append(1))
    assert len(built) == 2


def test_engine_key_follows_nested_module_replacement_and_hooks_route_away():
    """replacing a NESTED sub-module (model.l5.gcn1) must re-collect the watched tensors, and a model with
    forward hooks is not served by the engine (the hooks would not fire)."""
    m = M.Model(**MODEL_CASES[1][1]).eval()
    eng = f2.FusedEval(m)
    k0 = eng._state_key()
    old = m.l5.gcn1
    m.l5.gcn1 = M.unit_gcn(64, 128, old.PA.detach().numpy())
    k1 = eng._state_key()
    assert k1 != k0 and any(p is q for p in eng._watch for q in m.l5.gcn1.parameters())
    assert not any(p is q for p in eng._watch for q in old.parameters())


def test_unsupported_rel_channels_fail_at_construction():
    """CTRGC(72, 64) has 9 rel-channels, which the kernels are not built for: NotImplementedError when the
    module is BUILT, with the supported set in the message (it used to surface as a RuntimeError in the first forward)."""
    with pytest.raises(NotImplementedError, match='rel_channels'):
        M.CTRGC(72, 64)
    M.CTRGC(64, 64); M.CTRGC(3, 64); M.CTRGC(256, 128)
