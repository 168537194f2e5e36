"""-m gpu: full models.ctrgcn.Model on the HIP path against golden vectors from the
reference.

Forward bars are strict (north_star / SURVEY.md §8d): logits within 1e-3 absolute, top-1 indices
identical, CE loss, eval logits, extract_feature.

Gradient bars.  Every kernel reproduces its fp64 value to fp32 rounding on these very
activations (profiles/r01_parity_diagnostics/block_report.log: 3e-7 relative per block, fwd and
bwd).  End-to-end gradients of a ReLU network are nevertheless *discontinuous*: a pre-activation
that is exactly +4.6e-7 in fp64 (ucla_t64, l2, max-pool branch; unit_internal.log) lands on the
other side of zero in any fp32 evaluation that rounds differently, the ReLU mask flips and a
gradient entry worth 10 % of max|grad| appears or disappears; BatchNorm backward then spreads it
over the channel.  The reference itself shows the same events between its fp32 and fp64 runs
(fixture: *64 arrays).  So gradients are checked with flip-robust metrics against the fp64
reference: relative L2 error <= 5e-2 and cosine similarity >= 0.999.  Until round 3 the mildest case (ucla_t13) carried
ten times tighter bars; it was mild by luck: its fp64 forward has a ReLU input 4.65e-6 from zero at l10's OUTPUT (T = 4
there: 160 positions per channel), the HIP forward's error at that depth is 2.7e-6 of max|.| ~ 10, and with round 4's
temporal-branch kernels (csrc/tconv.hip, another summation order) that one mask lands on the other side: l10's input
gradient moves by 23 % of its max-norm at that position and dx by 1.1e-2, while every block of the case, fed the exact
inputs, reproduces its fp64 result to 7e-7 and the generic kernels on the same inputs reproduce dx to 6e-6
(tools/flip_report.py -> profiles/r04_flip_report_ucla_t13.txt).  The case now takes the flip-robust bars like the others and
its shapes (T = 13 / 7 / 4) joined the teacher-forced strict test instead (tests/test_gpu_blocks.py, BLOCK_CASES)
(tools/block_report.py, tools/dx_report.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cases import MODEL_CASES, MODEL_PARAM_SEED, MODEL_X_SEED, MODEL_LABEL_SEED   # noqa: E402
from params import fill_state_, make_input, make_labels, digest                  # noqa: E402
from tam_gcn_amd.models import ctrgcn as M                                         # noqa: E402

NOISE_K = 10.0


STRICT_CASES = ()                     # see the docstring: ucla_t13 was strict until round 4


def _check(name, got, ref32, ref64, rel, atol=0.0, strict=True):
    got = np.asarray(got, dtype=np.float64)
    noise = np.abs(np.asarray(ref32, dtype=np.float64) - ref64).max()
    scale = np.abs(ref64).max()
    diff = np.abs(got - ref64)
    if diff.size < 4 and not strict:
        # a scalar (unit_gcn.alpha) is ONE heavily cancelling sum: a single ReLU-mask flip moves it by ~1e-3 absolute, in
        # the reference's own fp32-vs-fp64 runs too (fixture: ucla_t52 l1.gcn1.alpha 0.0141 vs 0.0152) -- bound it by 25 %
        assert diff.max() <= 0.25 * scale + NOISE_K * noise + atol, f'{name}: {got} vs {ref64}'
        return
    l2 = np.sqrt((diff ** 2).sum()) / (np.sqrt((ref64 ** 2).sum()) + 1e-30)
    if strict:
        tol = rel * scale + NOISE_K * noise + atol
        over = float((diff > tol).mean()) if diff.size >= 100 else 0.0      # tiny tensors: the 5 x tol bound below
        assert over <= 0.05, (f'{name}: {over:.2%} of the entries exceed {tol:.3e} (max err {diff.max():.3e}, scale {scale:.3e}, '
                              f'ref noise {noise:.3e})')
        assert diff.max() <= 5 * tol, f'{name}: err {diff.max():.3e} > 5 x tol {tol:.3e}'
        assert l2 <= 5e-3 or diff.size < 100, f'{name}: relative L2 error {l2:.3e} (strict case)'
    assert l2 <= 5e-2, f'{name}: relative L2 error {l2:.3e}'
    cos = float((got * ref64).sum() / (np.sqrt((got ** 2).sum() * (ref64 ** 2).sum()) + 1e-30))
    assert cos >= 0.999, f'{name}: cosine similarity {cos:.5f}'


@pytest.mark.parametrize('case', MODEL_CASES, ids=lambda c: c[0])
def test_model_parity(case, golden_models):
    tag, margs, shape = case
    gold = golden_models
    dev = torch.device('cuda:0')
    m = M.Model(**margs)
    fill_state_(m.state_dict(), seed=MODEL_PARAM_SEED)
    m = m.to(dev).train()
    x = make_input(shape, seed=MODEL_X_SEED).to(dev).requires_grad_(True)
    lab = make_labels(shape[0], margs['num_class'], seed=MODEL_LABEL_SEED).to(dev)
    logits = m(x)
    loss = torch.nn.functional.cross_entropy(logits, lab)
    loss.backward()
    torch.cuda.synchronize()
    lg = logits.detach().cpu().numpy()
    ref = gold[f'{tag}/logits_train']
    assert np.abs(lg - ref).max() <= 1e-3, f'logits max-abs-diff {np.abs(lg - ref).max():.3e}'
    assert np.array_equal(lg.argmax(1), ref.argmax(1))                      # top-1 bit-exact
    assert abs(float(loss.detach()) - float(gold[f'{tag}/loss'])) <= 1e-3
    strict = tag in STRICT_CASES
    _check('dx', x.grad.cpu().numpy(), gold[f'{tag}/dx'], gold[f'{tag}/dx64'], 2e-3, strict=strict)
    gd32, gd64 = gold[f'{tag}/grad_digest'], gold[f'{tag}/grad_digest64']
    for i, (k, p) in enumerate(m.named_parameters()):
        assert k == str(gold[f'{tag}/param_keys'][i])
        g = digest(p.grad)
        n = p.numel()
        # digest = [sum, sum|.|, sum sq, head8, tail8]; compare the two sums against sum|.|
        for j, what in ((0, 'sum'), (1, 'abs-sum')):
            noise = abs(gd32[i][j] - gd64[i][j])
            rel = 1e-2 if strict else (0.25 if n == 1 else 5e-2)      # a scalar (alpha) is one heavily cancelling sum: flips move it most
            tol = rel * abs(gd64[i][1]) + NOISE_K * noise + 2e-6 * n
            assert abs(g[j] - gd64[i][j]) <= tol, f'{k}: {what} {g[j]} vs {gd64[i][j]} (tol {tol:.3e})'
        key = f'{tag}/grad/{k}'
        if key in gold.files:
            _check(k, p.grad.cpu().numpy(), gold[key], gold[f'{tag}/grad64/{k}'], 3e-3, 1e-6,
                   strict=strict or k.startswith('fc.'))     # fc grads depend on forward features only
    bd = gold[f'{tag}/buf_digest']
    for i, (k, b) in enumerate(m.named_buffers()):
        g = digest(b)
        assert abs(g[1] - bd[i][1]) <= 2e-4 * abs(bd[i][1]) + 1e-6, k
    # eval mode with the fixture's realistic running statistics
    sd = m.state_dict()
    with torch.no_grad():
        for k in sd:
            if 'running_' in k:
                sd[k].copy_(torch.from_numpy(gold[f'{tag}/evalbuf/{k}']))
    m.eval()
    with torch.no_grad():
        le = m(x.detach())
        f1, f2 = m.extract_feature(x.detach())
    le = le.cpu().numpy()
    assert np.abs(le - gold[f'{tag}/logits_eval']).max() <= 1e-3
    assert np.array_equal(le.argmax(1), gold[f'{tag}/logits_eval'].argmax(1))
    assert list(f1.shape) == list(gold[f'{tag}/feat_shape']) and f1 is f2
    fd = digest(f1)
    assert abs(fd[1] - gold[f'{tag}/feat_digest'][1]) <= 1e-4 * abs(gold[f'{tag}/feat_digest'][1])
    if shape[-1] == 1:
        x3 = x.detach()[..., 0].permute(0, 2, 3, 1).contiguous().view(shape[0], shape[2], -1)
        with torch.no_grad():
            l3 = m(x3).cpu().numpy()
        assert np.abs(l3 - gold[f'{tag}/logits_eval_3d']).max() <= 1e-3


def test_ntu_full_length_against_oracle():
    """BASELINE config 3 at its real clip length: NTU-RGB+D graph, 25 joints, 300 frames, 2 persons (one clip; the CPU
    oracle needs a few seconds).  Train-mode logits, loss and the fc / data_bn gradients (eval mode is pinned by the golden model cases: with this
    test's arbitrary running statistics it is ill-conditioned, logits ~1e5)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import ctrgcn_oracle as O
    margs = dict(num_class=60, num_point=25, num_person=2, graph='graph.ntu_rgb_d.Graph', graph_args=dict(labeling_mode='spatial'))
    shape = (1, 3, 300, 25, 2)
    dev = torch.device('cuda:0')
    m = M.Model(**margs)
    fill_state_(m.state_dict(), seed=MODEL_PARAM_SEED)
    x = make_input(shape, seed=MODEL_X_SEED)
    lab = make_labels(shape[0], 60, seed=MODEL_LABEL_SEED)
    sd = O.clone_state(m.state_dict(), requires_grad=True)            # CPU oracle on the same parameters
    ref = O.model_forward(x, sd, 25, training=True)
    torch.nn.functional.cross_entropy(ref, lab).backward()
    m = m.to(dev).train()
    out = m(x.to(dev))
    loss = torch.nn.functional.cross_entropy(out, lab.to(dev))
    loss.backward()
    assert (out.detach().cpu() - ref.detach()).abs().max() <= 1e-3
    assert torch.equal(out.argmax(1).cpu(), ref.argmax(1))
    for k in ('fc.weight', 'fc.bias'):                                  # depend on the forward features only
        g, r = dict(m.named_parameters())[k].grad.cpu(), sd[k].grad
        assert (g - r).abs().max() <= 2e-3 * r.abs().max() + 1e-6, k
    g, r = m.data_bn.weight.grad.cpu().double(), sd['data_bn.weight'].grad.double()   # through all ten blocks: flip-robust bar
    assert float((g - r).norm() / r.norm()) <= 5e-2 and float((g * r).sum() / (g.norm() * r.norm())) >= 0.999


@pytest.mark.parametrize('fix', ['sgd3b', 'sgd3s', 'sgd3'], ids=['lr0.01_b64', 'lr0.01_b4_first_steps', 'lr0.05_b4_first_steps'])
@pytest.mark.parametrize('flat', [False, True], ids=['torch.optim.SGD', 'ParamArena+SGDNesterov'])
def test_harness_sgd_steps(flat, fix, golden_models):
    """SURVEY §8c-ii on the HIP path: three steps of the harness recipe against the losses and final state captured
    from the reference model -- once with the stock optimiser the reference's processor builds (drop-in), once with the
    flat parameter arena / gradient bucket / SGDNesterov the data-parallel step uses.

    'sgd3b' (lr 0.01, 64 clips x 32 frames): losses and every tensor of the final state (parameters, running
    statistics) within 1e-3 (+ the reference's own fp32-vs-fp64 noise).  The two 4-clip fixtures have 1040 positions per
    channel and are chaotic at fp32 resolution: the fp64 trajectory has ReLU inputs within 1e-6 of zero at every step, ONE mask
    of the first forward differs on the HIP path (shown, not assumed: tools/sgd_fixture_report.py ->
    profiles/r03_sgd_fixture_report.txt) and the trajectories separate from there.  They pin the first loss (a pure forward,
    1e-4), the second and third losses and every state tensor within about twice the measured deviations;
    tests/test_gpu_blocks.py holds every block's gradients to 1e-5 once flips are excluded."""
    from cases import SGD_CASES
    from tam_gcn_amd.distributed import ParamArena, SGDNesterov
    lr, nb, nt = SGD_CASES[fix]
    dev = torch.device('cuda:0')
    m = M.Model(**MODEL_CASES[0][1])
    fill_state_(m.state_dict(), seed=43)
    m = m.to(dev).train()
    if flat:
        arena = ParamArena(m)
        bucket = arena.grad_bucket()
        opt = SGDNesterov(arena.params, lr=lr, momentum=0.9, weight_decay=1e-4, arena=arena, bucket=bucket)
    else:
        opt = torch.optim.SGD(m.parameters(), lr=lr, momentum=0.9, nesterov=True, weight_decay=1e-4)
    losses = []
    for step in range(3):
        x = make_input((nb, 3, nt, 20, 1), seed=100 + step).to(dev)
        lab = make_labels(nb, 10, seed=200 + step).to(dev)
        if flat:
            bucket.zero()
        else:
            opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(m(x), lab)
        loss.backward()
        if flat:
            bucket.pack()
        opt.step()
        losses.append(float(loss.detach()))
    ref, ref64 = golden_models[f'{fix}/losses'], golden_models[f'{fix}/losses64']
    # Measured deviations from the fp64 reference on MI355X, identical in the exact-fp32 and the split mode.  Cause on record:
    # ONE ReLU mask of the first forward (l4's output, fp64 pre-activation 1.4e-6) differs from fp64; from there the
    # trajectories separate, and WHERE they go depends on every kernel's summation order:
    #   round 3 (profiles/r03_sgd_fixture_report.txt): loss[1] 2.1e-3 (sgd3s) / 1.4e-3 (sgd3), loss[2] 5.4e-2 / 2.0e-1, state
    #     tensors of >= 256 elements 2.3e-2 / 1.2e-1;
    #   round 4, temporal branches on csrc/tconv.hip (profiles/r04_sgd_fixture_report.txt): the same single mask differs, loss[1]
    #     4.4e-3 / 1.7e-3, loss[2] 1.2e-1 / 6.1e-2, large tensors 2.7e-2 / 3.5e-2 -- the two fixtures swapped places.
    # Bounds = the larger of the two rounds' figures x ~2 for both fixtures (the chaos is the fixture's, not one recipe's).
    LOSS1 = {'sgd3s': 1e-2, 'sgd3': 1e-2}
    LOSS2 = {'sgd3s': 4e-1, 'sgd3': 4e-1}
    BIG = {'sgd3s': 2.5e-1, 'sgd3': 2.5e-1}
    if fix == 'sgd3b':
        ltol = 1e-3 * np.abs(ref64) + NOISE_K * np.abs(ref - ref64)
        assert (np.abs(np.array(losses) - ref64) <= ltol).all(), (losses, ref, ref64)
    else:
        assert abs(losses[1] - ref64[1]) <= LOSS1[fix] * abs(ref64[1]), (losses, ref64)
        assert abs(losses[2] - ref64[2]) <= LOSS2[fix] * abs(ref64[2]), (losses, ref64)
        assert np.isfinite(losses).all()
    assert abs(losses[0] - ref[0]) <= 1e-4                                       # the first loss is a pure forward
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in golden_models[f'{fix}/keys']]
    got = np.stack([digest(v) for v in sd.values()])
    refd, refd64 = golden_models[f'{fix}/state_digest'], golden_models[f'{fix}/state_digest64']
    noise = np.abs(refd[:, 1] - refd64[:, 1])           # the reference's own fp32-vs-fp64 difference of every state tensor
    small = np.array([v.numel() < 256 for v in sd.values()])
    if fix == 'sgd3b':
        # sum |.| of every state tensor within 1e-3 of the fp64 reference, plus NOISE_K x the reference's own fp32 noise on
        # that tensor.  Tensors of fewer than 256 elements (conv1/conv2 biases, which enter only through p_u - q_v, alpha,
        # BatchNorm affines of 16 channels) get 2e-2: their gradients are heavily cancelling sums that a single ReLU-mask
        # flip anywhere upstream moves by per cent (the reference's own two precisions differ by up to 2.3 % on them)
        bad = np.abs(got[:, 1] - refd64[:, 1]) > np.where(small, 2e-2, 1e-3) * np.abs(refd64[:, 1]) + NOISE_K * noise + 2e-4
    else:
        # chaotic fixtures (see above): tensors of >= 256 elements within BIG, the cancelling-sum tensors below that size
        # within 200 % (measured up to 161 %, r04 report; the oracle, bit-compatible arithmetic, reproduces both fixtures to 2e-4:
        # tests/test_model_cpu.py)
        bad = np.abs(got[:, 1] - refd64[:, 1]) > np.where(small, 2.0, BIG[fix]) * np.abs(refd64[:, 1]) + 1e-2
        assert np.isfinite(got).all()
    assert not bad.any(), [(k, got[i, 1], refd[i, 1], refd64[i, 1]) for i, k in enumerate(sd.keys()) if bad[i]][:5]


def test_frame_means_handed_from_block_to_block(golden_models):
    """Blocks l1..l9 leave the frame means of their output beside it (ops.add_act_fwd(xbar=True)) and the next block's CTRGC
    takes them instead of launching tamgcn_tmean: same logits and gradients to fp32 rounding as with the hand-over off, one
    tmean launch left (l1's, on the stem output), and a tensor written in place between two blocks is NOT trusted."""
    from tam_gcn_amd import ops, functional as Fn
    dev = torch.device('cuda:0')
    tag, margs, shape = MODEL_CASES[2]
    m = M.Model(**margs)
    fill_state_(m.state_dict(), seed=MODEL_PARAM_SEED)
    m = m.to(dev).train()
    x = make_input(shape, seed=MODEL_X_SEED).to(dev)
    lab = make_labels(shape[0], margs['num_class'], seed=MODEL_LABEL_SEED).to(dev)
    calls = []
    orig = ops.tmean

    def spy(*a, **k):
        calls.append(1)
        return orig(*a, **k)
    res = {}
    try:
        ops.tmean = spy
        for fold in (True, False):
            ops.XBAR_FOLD = fold
            del calls[:]
            for p in m.parameters():
                p.grad = None
            logits = m(x)
            torch.nn.functional.cross_entropy(logits, lab).backward()
            torch.cuda.synchronize()
            res[fold] = (logits.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}, len(calls))
    finally:
        ops.tmean, ops.XBAR_FOLD = orig, True
    assert res[True][2] == 1 and res[False][2] == 10
    assert float((res[True][0] - res[False][0]).abs().max()) <= 1e-5 * float(res[False][0].abs().max())
    # gradients: the two runs differ by the rounding of x-bar (1e-7), which can put a ReLU input of this 4-clip case on the other
    # side (see the module docstring): the flip-robust bars of test_model_parity, 25 % for the scalar alpha
    for k, g in res[False][1].items():
        bar = 0.25 if g.numel() == 1 else 5e-2
        assert float((res[True][1][k] - g).norm()) <= bar * float(g.norm()) + 1e-7, k
    # an in-place write between two blocks invalidates the hand-over
    out = m.l1(Fn.StemFn.run(m.data_bn, x.unsqueeze(-1) if x.dim() == 4 else x, m.data_bn.weight, m.data_bn.bias) if False else
               torch.rand(2, 3, 16, 20, device=dev))
    assert Fn.taken_xbar(out, out.shape[1]) is not None
    with torch.no_grad():
        out.mul_(2.0)
    assert Fn.taken_xbar(out, out.shape[1]) is None


def test_frozen_backbone_usage():
    """The cross-modal caller freezes the CTR-GCN and reads extract_feature (reference models/resnet_gcn_attention.py:24-26,
    82-85): frozen parameters get no gradients, a downstream head still trains, and a gradient w.r.t. the input equals the
    one of the unfrozen model."""
    dev = torch.device('cuda:0')
    m = M.Model(**MODEL_CASES[0][1])
    fill_state_(m.state_dict(), seed=MODEL_PARAM_SEED)
    m = m.to(dev).eval()
    x = make_input((2, 3, 52, 20, 1), seed=MODEL_X_SEED).to(dev)
    xr = x.clone().requires_grad_(True)
    f_ref, _ = m.extract_feature(xr)
    assert tuple(f_ref.shape) == (2, 256, 13, 20, 1)                      # the shape comment at resnet_gcn_attention.py:81
    f_ref.square().mean().backward()
    g_unfrozen = xr.grad.clone()
    for p in m.parameters():
        p.requires_grad = False
        p.grad = None
    head = torch.nn.Linear(256, 4).to(dev)
    f, f2 = m.extract_feature(x)                                           # input without grad: nothing to differentiate upstream
    assert f is f2 and not f.requires_grad
    out = head(f.mean((2, 3, 4)))
    out.sum().backward()
    assert head.weight.grad is not None and all(p.grad is None for p in m.parameters())
    xr2 = x.clone().requires_grad_(True)
    f3, _ = m.extract_feature(xr2)
    f3.square().mean().backward()
    assert all(p.grad is None for p in m.parameters())
    assert torch.allclose(xr2.grad, g_unfrozen, rtol=1e-4, atol=1e-7)


def test_dataparallel_wrapper_single_device():
    """The harness wraps the model in nn.DataParallel (reference processor/io.py:86-87); on one device that is a plain call."""
    dev = torch.device('cuda:0')
    m = M.Model(**MODEL_CASES[0][1])
    fill_state_(m.state_dict(), seed=MODEL_PARAM_SEED)
    m = m.to(dev).eval()
    x = make_input((2, 3, 13, 20, 1), seed=MODEL_X_SEED).to(dev)
    dp = torch.nn.DataParallel(m, device_ids=[0])
    with torch.no_grad():
        assert torch.equal(dp(x), m(x))
    assert [k for k in dp.state_dict()][0].startswith('module.')


def test_full_size_properties():
    """BASELINE configs[1] at its full size (256 clips x 3 x 64 x 20): no oracle run is affordable here, so the properties
    the domain offers.  (1) eval mode: a clip's logits do not depend on its batch (full grids, XCD mapping, every tile path
    vs a 4-clip launch); (2) train mode: permuting the batch permutes the logits (BatchNorm statistics are order-free up to
    summation order) and leaves the summed loss gradient of a parameter unchanged."""
    dev = torch.device('cuda:0')
    m = M.Model(**MODEL_CASES[0][1])
    fill_state_(m.state_dict(), seed=MODEL_PARAM_SEED)
    m = m.to(dev)
    g = torch.Generator().manual_seed(11)
    x = (torch.rand(256, 3, 64, 20, 1, generator=g) * 2 - 1).to(dev)
    lab = torch.randint(0, 10, (256,), generator=g).to(dev)
    m.train()                                            # one step with momentum 1: running statistics := batch statistics
    for mod in m.modules():
        if isinstance(mod, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            mod.momentum = 1.0
    with torch.no_grad():
        m(x)
    m.eval()
    with torch.no_grad():
        full = m(x)
        sub = m(x[100:104].contiguous())
    assert torch.isfinite(full).all()
    assert (full[100:104] - sub).abs().max() <= 1e-4 * full.abs().max()
    m.train()
    perm = torch.randperm(256, generator=g).to(dev)
    out = m(x)
    torch.nn.functional.cross_entropy(out, lab, reduction='sum').backward()
    g1 = m.l10.tcn1.branches[0][3].conv.weight.grad.clone()
    for p in m.parameters():
        p.grad = None
    outp = m(x[perm].contiguous())
    torch.nn.functional.cross_entropy(outp, lab[perm], reduction='sum').backward()
    g2 = m.l10.tcn1.branches[0][3].conv.weight.grad
    assert (outp - out[perm]).abs().max() <= 2e-4 * out.abs().max()
    assert (g1 - g2).norm() <= 2e-3 * g1.norm()


def test_ntu_full_size_properties():
    """BASELINE configs[3] at its stated size: NTU-RGB+D graph, 25 joints x 300 frames x 2 persons, batch 128 (N*M = 256 skeleton
    sequences, 56 GiB of saved activations).  No oracle run is affordable at this size, so the domain's properties: (1) eval mode: a
    clip's logits do not depend on the batch it sits in (128-clip launch vs a 2-clip launch: every tile path, the clip-chunked
    launches, the V = 25 line buffers with their 28-float frame pitch); (2) train mode: permuting the batch permutes the logits and
    leaves a summed-loss parameter gradient unchanged."""
    dev = torch.device('cuda:0')
    margs = dict(num_class=60, num_point=25, num_person=2, graph='graph.ntu_rgb_d.Graph', graph_args=dict(labeling_mode='spatial'))
    m = M.Model(**margs)
    fill_state_(m.state_dict(), seed=MODEL_PARAM_SEED)
    m = m.to(dev)
    g = torch.Generator().manual_seed(12)
    B = 128
    x = (torch.rand(B, 3, 300, 25, 2, generator=g) * 2 - 1).to(dev)
    lab = torch.randint(0, 60, (B,), generator=g).to(dev)
    m.train()
    for mod in m.modules():
        if isinstance(mod, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            mod.momentum = 1.0                            # running statistics := this batch's
    with torch.no_grad():
        m(x)
    m.eval()
    with torch.no_grad():
        full = m(x)
        sub = m(x[60:62].contiguous())
    assert torch.isfinite(full).all()
    assert (full[60:62] - sub).abs().max() <= 1e-4 * full.abs().max()
    m.train()
    perm = torch.randperm(B, generator=g).to(dev)
    out = m(x)
    torch.nn.functional.cross_entropy(out, lab, reduction='sum').backward()
    g1 = m.l10.tcn1.branches[0][3].conv.weight.grad.clone()
    g1b = m.l2.gcn1.convs[1].conv3.weight.grad.clone()
    for p in m.parameters():
        p.grad = None
    outp = m(x[perm].contiguous())
    torch.nn.functional.cross_entropy(outp, lab[perm], reduction='sum').backward()
    assert (outp - out[perm]).abs().max() <= 2e-4 * out.abs().max()
    assert (g1 - m.l10.tcn1.branches[0][3].conv.weight.grad).norm() <= 2e-3 * g1.norm()
    assert (g1b - m.l2.gcn1.convs[1].conv3.weight.grad).norm() <= 5e-3 * g1b.norm()


def test_eval_fused_path_and_coefficient_cache(monkeypatch):
    """Row f2: under no_grad in eval mode the blocks run the fused form (cached BatchNorm coefficients, MS-TCN branches that
    finish relu(bn(.) + residual) in their own epilogues).  It must equal the unfused pipeline, follow a change of the running
    statistics made by a training step (whose kernel writes them through raw pointers) and of the affine parameters, and
    launch fewer kernels."""
    from tam_gcn_amd import functional as Fn, _lib
    monkeypatch.setenv('TAMGCN_F2', '0')                   # this test is about the GENERAL eval path (any batch size, any V);
    dev = torch.device('cuda:0')                           # batches this small otherwise go to tam_gcn_amd.f2 (tests/test_gpu_f2.py)
    m = M.Model(**MODEL_CASES[0][1])
    fill_state_(m.state_dict(), seed=MODEL_PARAM_SEED)
    m = m.to(dev).eval()
    x = make_input((3, 3, 52, 20, 1), seed=MODEL_X_SEED).to(dev)

    def both():
        with torch.no_grad():
            monkeypatch.setattr(Fn, 'EVAL_FUSED', True)
            a, (fa, _) = m(x), m.extract_feature(x)
            monkeypatch.setattr(Fn, 'EVAL_FUSED', False)
            b, (fb, _) = m(x), m.extract_feature(x)
        monkeypatch.setattr(Fn, 'EVAL_FUSED', True)
        return a, b, fa, fb

    a, b, fa, fb = both()
    assert (a - b).abs().max() <= 2e-5 * b.abs().max() and (fa - fb).abs().max() <= 2e-5 * fb.abs().max()
    m.train()                                              # a training step moves every running statistic
    m(make_input((4, 3, 52, 20, 1), seed=5).to(dev) * 3 + 0.5).sum().backward()
    m.eval()
    a2, b2, _, _ = both()
    assert (a2 - b2).abs().max() <= 2e-5 * b2.abs().max()
    assert (a2 - a).abs().max() > 1e-3 * a.abs().max()      # the statistics did change: a stale cache would reproduce `a`
    with torch.no_grad():
        m.l3.tcn1.branches[0][1].weight.mul_(1.5)          # in-place parameter update (an optimiser step, load_state_dict)
    a3, b3, _, _ = both()
    assert (a3 - b3).abs().max() <= 2e-5 * b3.abs().max() and (a3 - a2).abs().max() > 0

    class Count:                                           # ABI launches of one forward
        def __init__(self, lib):
            self.lib, self.n = lib, 0

        def __getattr__(self, name):
            fn = getattr(self.lib, name)
            if not name.startswith('tamgcn_') or name in ('tamgcn_last_error', 'tamgcn_conv_nparts', 'tamgcn_wgrad_max_split',
                                                         'tamgcn_ctrgc_tiled_supported', 'tamgcn_ctrgc_tiled_chunks'):
                return fn

            def w(*args):
                self.n += 1
                return fn(*args)
            return w

    real = _lib.load()
    counts = []
    for fused in (True, False):
        cnt = Count(real)
        monkeypatch.setattr(_lib, '_lib', cnt)
        monkeypatch.setattr(Fn, 'EVAL_FUSED', fused)
        with torch.no_grad():
            m(x)
        counts.append(cnt.n)
    monkeypatch.setattr(_lib, '_lib', real)
    assert counts[0] <= 125 and counts[0] <= counts[1] - 40, counts


def test_graphed_forward_replays_the_eval_path(golden_models):
    """tam_gcn_amd.inference.GraphedForward: the HIP-graph replay of the eval forward returns the eager logits bit for bit,
    for several inputs and two batch shapes, follows a parameter change after reset(), and extract_feature works too."""
    from tam_gcn_amd.inference import GraphedForward
    tag, margs, shape = next(c for c in MODEL_CASES if c[0] == 'ucla_t64')
    dev = torch.device('cuda:0')
    m = M.Model(**margs)
    fill_state_(m.state_dict(), seed=MODEL_PARAM_SEED)
    m = m.to(dev)
    with pytest.raises(ValueError):
        GraphedForward(m)                                   # still in train mode
    m.eval()
    fast = GraphedForward(m)
    for seed in (1, 2, 3):
        for nb in (shape[0], 1):
            x = make_input((nb,) + tuple(shape[1:]), seed=seed).to(dev)
            with torch.no_grad():
                ref = m(x)
            got = fast(x).clone()
            assert torch.equal(got, ref), (seed, nb, float((got - ref).abs().max()))
    assert len(fast._graphs) == 2
    with torch.no_grad():
        m.fc.bias.add_(1.0)
    fast.reset()
    x = make_input(shape, seed=1).to(dev)
    with torch.no_grad():
        assert torch.equal(fast(x), m(x))
    feat = GraphedForward(m, method='extract_feature')
    with torch.no_grad():
        a, b = m.extract_feature(x)
    fa, fb = feat(x)
    assert torch.equal(fa, a) and torch.equal(fb, b)


def test_backward_through_eval_mode_matches_the_oracle(golden_models):
    """Gradients through the model in eval() mode (running statistics, no updates) -- the cross-modal caller fine-tunes a
    head on top of the backbone's features with autograd on (models/resnet_gcn_attention.py:82-85): the input gradient and
    every parameter gradient against the CPU oracle evaluated in fp64 with the fixture's running statistics, and the
    buffers must stay untouched."""
    from oracle import ctrgcn_oracle as O
    tag, margs, shape = next(c for c in MODEL_CASES if c[0] == 'ucla_t13')
    dev = torch.device('cuda:0')
    m = M.Model(**margs)
    fill_state_(m.state_dict(), seed=MODEL_PARAM_SEED)
    sd0 = m.state_dict()
    with torch.no_grad():
        for k in sd0:
            if 'running_' in k:
                sd0[k].copy_(torch.from_numpy(golden_models[f'{tag}/evalbuf/{k}']))
    sd = {k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in m.state_dict().items()}
    pkeys = [k for k, _ in m.named_parameters()]
    for k in pkeys:
        sd[k].requires_grad_(True)
    xo = make_input(shape, seed=MODEL_X_SEED).double().requires_grad_(True)
    cot = make_input((shape[0], margs['num_class']), seed=77).double()
    lo = O.model_forward(xo, sd, margs['num_point'], training=False)
    (lo * cot).sum().backward()
    m = m.to(dev).eval()
    before = {k: b.detach().clone() for k, b in m.named_buffers()}
    x = make_input(shape, seed=MODEL_X_SEED).to(dev).requires_grad_(True)
    lg = m(x)
    (lg * cot.float().to(dev)).sum().backward()
    torch.cuda.synchronize()
    rel = lambda a, b: float((a.detach().cpu().double() - b).abs().max() / (b.abs().max() + 1e-30))   # noqa: E731
    assert rel(lg, lo.detach()) <= 1e-5

    # Flip-robust bars, as for the end-to-end gradients in train mode (module docstring): relative L2 <= 2e-2, cosine >= 0.9995;
    # fc (forward features only) tight.  The fp64 evaluation has ReLU inputs within 1e-6 of zero (tools/sgd_fixture_report.py
    # counts 2 per 2.2e6 on the sibling 4-clip fixture): which side an fp32 evaluation puts them on depends on its summation
    # order.  The VALU aggregation of rounds 1-2 agreed with fp64 on all of them and this test held 2e-4 in max-norm; the
    # matrix-core aggregation of round 3 flips one mask at l9's output, where a clip is 256 x 4 x 20 values: every gradient
    # below it moves by 5e-3..2e-2 in relative L2 (tools/eval_bwd_report.py: l10 and fc at 7e-6, l9..l1 at 6e-3..1.7e-2,
    # the scalar alphas up to 1.3e-1) while the train-mode run of the same case has no flip (every tensor <= 6e-5).  The strict
    # statement about eval-mode gradients is the teacher-forced block test (tests/test_gpu_blocks.py, bn = 'eval': 1e-5).
    def close(name, a, b, l2max, tiny):
        a, b = a.detach().cpu().double(), b.detach().double()
        if a.numel() < 4:
            assert float((a - b).abs().max()) <= tiny * float(b.abs().max()) + 1e-12, f'{name}: {a} vs {b}'
            return
        l2 = float((a - b).norm() / (b.norm() + 1e-30))
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
        assert l2 <= l2max and cos >= 0.9995, f'{name}: relative L2 {l2:.2e}, cosine {cos:.5f}'

    close('dx', x.grad, xo.grad, 2e-2, 0.25)
    for k, p in m.named_parameters():
        ref = sd[k].grad
        assert p.grad is not None, k
        if float(ref.abs().max()) < 1e-12:
            assert float(p.grad.abs().max()) < 1e-6, k
            continue
        close(k, p.grad, ref, 1e-4 if k.startswith('fc.') else 3e-2, 0.25)
    for k, b in m.named_buffers():
        assert torch.equal(b, before[k]), f'{k} changed in eval mode'


def test_non_adaptive_graph_and_dropout_head():
    """Model(adaptive=False): the adjacency is a constant of each unit_gcn, not a parameter (reference models/ctrgcn.py:225-228,
    248-251) -- state dict without PA keys, logits / input gradient / parameter gradients against the oracle fed the same
    constant; drop_out > 0 takes the reference's own pooling + nn.Dropout + fc route (:343-348): identical to the fused
    head in eval mode, runs and back-propagates in train mode."""
    from oracle import ctrgcn_oracle as O
    margs = dict(MODEL_CASES[0][1], adaptive=False)
    shape = (3, 3, 16, 20, 1)
    dev = torch.device('cuda:0')
    m = M.Model(**margs)
    fill_state_(m.state_dict(), seed=MODEL_PARAM_SEED)
    assert not any(k.endswith('PA') for k in m.state_dict()) and len(m.state_dict()) == 892 - 10
    sd = {k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in m.state_dict().items()}
    for i in range(1, 11):
        sd[f'l{i}.gcn1.PA'] = getattr(m, f'l{i}').gcn1.A.double().clone()
    for k, _ in m.named_parameters():
        sd[k].requires_grad_(True)
    xo = make_input(shape, seed=5).double().requires_grad_(True)
    lab = make_labels(shape[0], margs['num_class'], seed=6)
    lo = O.model_forward(xo, sd, margs['num_point'], training=True)
    torch.nn.functional.cross_entropy(lo, lab).backward()
    m = m.to(dev).train()
    x = make_input(shape, seed=5).to(dev).requires_grad_(True)
    lg = m(x)
    torch.nn.functional.cross_entropy(lg, lab.to(dev)).backward()
    torch.cuda.synchronize()
    rel = lambda a, b: float((a.detach().cpu().double() - b.detach()).abs().max() / (b.detach().abs().max() + 1e-30))   # noqa: E731
    assert rel(lg, lo) <= 1e-4
    assert rel(m.fc.weight.grad, sd['fc.weight'].grad) <= 1e-3
    assert rel(m.l1.gcn1.convs[0].conv3.weight.grad, sd['l1.gcn1.convs.0.conv3.weight'].grad) <= 5e-2    # through ten blocks of ReLU masks
    assert getattr(m.l1.gcn1, 'A').device.type == 'cuda' and not m.l1.gcn1.A.requires_grad
    # drop_out > 0: torch's own head
    md = M.Model(**dict(MODEL_CASES[0][1], drop_out=0.5))
    fill_state_(md.state_dict(), seed=MODEL_PARAM_SEED)
    m0 = M.Model(**MODEL_CASES[0][1])
    m0.load_state_dict(md.state_dict())
    md, m0 = md.to(dev).eval(), m0.to(dev).eval()
    xe = make_input(shape, seed=7).to(dev)
    with torch.no_grad():
        a, b = md(xe), m0(xe)
    assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-6
    md.train()
    xt = make_input(shape, seed=8).to(dev).requires_grad_(True)
    out = md(xt)
    out.sum().backward()
    assert out.shape == (shape[0], margs['num_class']) and torch.isfinite(xt.grad).all() and md.fc.weight.grad is not None
