"""Model-level CPU tests: init parity with the reference, state-dict ABI, oracle
vs golden logits / gradients / running stats, SGD-step contract, fail-loud on CPU."""
import numpy as np
import pytest
import torch

from cases import MODEL_CASES, MODEL_PARAM_SEED, MODEL_X_SEED, MODEL_LABEL_SEED, MODEL_INIT_SEED
from params import fill_state_, make_input, make_labels, digest
from tam_gcn_amd.models import ctrgcn as M
from oracle import ctrgcn_oracle as O


def _args(margs):
    a = dict(margs)
    return a


@pytest.mark.parametrize('case', MODEL_CASES[:1] + MODEL_CASES[3:], ids=lambda c: c[0])
def test_init_bit_identical_to_reference(case, golden_models):
    """Same seed => same initial state-dict as the reference (a9: init helpers,
    RNG consumption order, degenerate defaults alpha=0 / bn=1e-6 / offset=0)."""
    tag, margs, _ = case
    torch.manual_seed(MODEL_INIT_SEED)
    m = M.Model(**margs)
    keys = list(golden_models[f'{tag}/keys'])
    assert list(m.state_dict().keys()) == keys
    ref = golden_models[f'{tag}/init_digest']
    got = np.stack([digest(v) for v in m.state_dict().values()])
    np.testing.assert_allclose(got, ref, rtol=0, atol=0)


def test_state_dict_abi():
    m = M.Model(num_class=10, num_point=20, num_person=1, graph='graph.ucla.Graph',
                graph_args=dict(labeling_mode='spatial'))
    sd = m.state_dict()
    assert len(sd) == 892
    assert sum(p.numel() for p in m.parameters()) == 1693260
    assert tuple(sd['l5.gcn1.PA'].shape) == (3, 20, 20)
    assert tuple(sd['l5.gcn1.convs.2.conv3.weight'].shape) == (128, 64, 1, 1)
    assert tuple(sd['l5.tcn1.branches.1.3.conv.weight'].shape) == (32, 32, 5, 1)
    assert 'l5.residual.conv.weight' in sd and 'l2.residual.conv.weight' not in sd
    assert 'l1.gcn1.down.0.weight' in sd and 'l2.gcn1.down.0.weight' not in sd
    # strict load incl. DataParallel-style 'module.' prefix stripping (torchlight/io.py:65-66)
    m2 = M.Model(num_class=10, num_point=20, num_person=1, graph='graph.ucla.Graph')
    pref = {'module.' + k: v for k, v in sd.items()}
    m2.load_state_dict({k[len('module.'):]: v for k, v in pref.items()}, strict=True)
    for p in m2.parameters():               # freezing as models/resnet_gcn_attention.py:24-26 does
        p.requires_grad = False


def test_graph_none_raises_valueerror():
    with pytest.raises(ValueError):
        M.Model(graph=None)


def test_mstcn_branch_assert():
    with pytest.raises(AssertionError):
        M.MultiScale_TemporalConv(64, 62, dilations=[1, 2])


def test_cpu_tensor_fails_loudly():
    m = M.Model(num_class=10, num_point=20, num_person=1, graph='graph.ucla.Graph')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m(torch.zeros(1, 3, 8, 20, 1))


@pytest.mark.parametrize('case', [MODEL_CASES[0], MODEL_CASES[3]], ids=lambda c: c[0])
def test_oracle_model_vs_golden(case, golden_models):
    tag, margs, shape = case
    gold = golden_models
    m = M.Model(**margs)
    fill_state_(m.state_dict(), seed=MODEL_PARAM_SEED)
    sd = O.clone_state(m.state_dict(), requires_grad=True)
    x = make_input(shape, seed=MODEL_X_SEED).requires_grad_(True)
    lab = make_labels(shape[0], margs['num_class'], seed=MODEL_LABEL_SEED)
    logits = O.model_forward(x, sd, margs['num_point'], training=True)
    loss = torch.nn.functional.cross_entropy(logits, lab)
    loss.backward()
    np.testing.assert_allclose(logits.detach().numpy(), gold[f'{tag}/logits_train'], rtol=1e-3, atol=1e-3)
    assert np.array_equal(logits.detach().numpy().argmax(1), gold[f'{tag}/logits_train'].argmax(1))
    np.testing.assert_allclose(float(loss.detach()), float(gold[f'{tag}/loss']), rtol=1e-4)
    np.testing.assert_allclose(x.grad.numpy(), gold[f'{tag}/dx'], rtol=5e-3, atol=1e-6)
    pkeys = list(gold[f'{tag}/param_keys'])
    gd = gold[f'{tag}/grad_digest']
    for i, k in enumerate(pkeys):
        g = digest(sd[k].grad)
        tol = 5e-3 * abs(gd[i][1]) + 1e-6
        assert abs(g[0] - gd[i][0]) <= tol and abs(g[1] - gd[i][1]) <= tol, k
    bkeys = list(gold[f'{tag}/buf_keys'])
    bd = gold[f'{tag}/buf_digest']
    for i, k in enumerate(bkeys):
        g = digest(sd[k])
        assert abs(g[1] - bd[i][1]) <= 1e-4 * abs(bd[i][1]) + 1e-6, k
    with torch.no_grad():
        sde = {k: v.detach() for k, v in sd.items()}
        for k in list(sde):
            if 'running_' in k:
                sde[k] = torch.from_numpy(gold[f'{tag}/evalbuf/{k}'])
        le = O.model_forward(x.detach(), sde, margs['num_point'], training=False)
        f1, _ = O.model_extract_feature(x.detach(), sde, margs['num_point'], training=False)
    np.testing.assert_allclose(le.numpy(), gold[f'{tag}/logits_eval'], rtol=1e-3, atol=1e-3)
    assert list(f1.shape) == list(gold[f'{tag}/feat_shape'])
    fd = digest(f1)
    assert abs(fd[1] - gold[f'{tag}/feat_digest'][1]) <= 1e-3 * abs(gold[f'{tag}/feat_digest'][1])
    if shape[-1] == 1:
        x3 = x.detach()[..., 0].permute(0, 2, 3, 1).contiguous().view(shape[0], shape[2], -1)
        with torch.no_grad():
            l3 = O.model_forward(x3, sde, margs['num_point'], training=False)
        np.testing.assert_allclose(l3.numpy(), gold[f'{tag}/logits_eval_3d'], rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize('fix', ['sgd3', 'sgd3s', 'sgd3b'])
def test_harness_sgd_steps_oracle(fix, golden_models):
    """three steps of the harness recipe (SGD momentum 0.9, nesterov, wd 1e-4 + CE, reference
    processor/recognition_rgb.py:19-28) captured from the reference model: the oracle under torch.optim.SGD reproduces
    the losses and the final state (parameters, running statistics, num_batches_tracked)."""
    from cases import SGD_CASES
    lr, nb, nt = SGD_CASES[fix]
    margs = MODEL_CASES[0][1]
    m = M.Model(**margs)
    fill_state_(m.state_dict(), seed=43)
    sd = O.clone_state(m.state_dict(), requires_grad=True)
    params = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.SGD(params, lr=lr, momentum=0.9, nesterov=True, weight_decay=1e-4)
    losses = []
    for step in range(3):
        x = make_input((nb, 3, nt, 20, 1), seed=100 + step)
        lab = make_labels(nb, 10, seed=200 + step)
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(O.model_forward(x, sd, 20, training=True), lab)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    np.testing.assert_allclose(losses, golden_models[f'{fix}/losses'], rtol=2e-5, atol=2e-6)
    assert list(sd.keys()) == [str(k) for k in golden_models[f'{fix}/keys']]
    got = np.stack([digest(v) for v in sd.values()])
    ref = golden_models[f'{fix}/state_digest']
    np.testing.assert_allclose(got[:, 1], ref[:, 1], rtol=2e-4, atol=1e-5)        # sum |.| of every state tensor
