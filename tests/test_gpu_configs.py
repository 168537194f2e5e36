"""-m gpu: the BASELINE.json configurations as WHOLE paths (each kernel family is pinned elsewhere; these tests
chain them the way the configurations do).

config 0  feeder -> (1, 3, 52, 20, 1) -> Model.forward -> (1, 10)            (reference feeder/feeder_nucla_gcn.py:85-130,154
                                                                              -> models/ctrgcn.py:324-348)
config 2  four Models on joint / bone / motion / bone-motion clips derived on the GPU, ONE parameter arena, ONE gradient
          bucket, one step: losses and the packed bucket against the oracle on the four derived inputs, flat SGD state
          against four per-model torch.optim.SGD                             (feeder_nucla_gcn.py:119-127, the harness recipe
                                                                              processor/recognition_rgb.py:19-28)
          + the same four models on four HIP streams, eager and captured into a HIP graph (bench.py --config 4stream)
config 4  TCN_GCN_unit(256, 256) on (clips, 256, 512, 64): size-independent properties at 32 clips x 512 frames
          (module parity at V = 64 over several frame chunks: tests/golden cases *_v64_t*)
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cases import MODEL_CASES, MODEL_PARAM_SEED                                  # noqa: E402
from params import fill_state_, make_input, make_labels                          # noqa: E402
from oracle import ctrgcn_oracle as O                                            # noqa: E402
from tam_gcn_amd import ops                                                      # noqa: E402
from tam_gcn_amd.models import ctrgcn as M                                       # noqa: E402
from tam_gcn_amd.feeder.feeder_nucla_gcn import Feeder, BONE_PARENT              # noqa: E402

UCLA = MODEL_CASES[0][1]
STREAMS = ('joint', 'bone', 'motion', 'bone_motion')
FEEDER_GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'feeder.npz'))


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


# ---------------------------------------------------------------------------------------------------------------------
# config 0
# ---------------------------------------------------------------------------------------------------------------------
def test_config0_feeder_sample_through_model(tmp_path, golden_models):
    """One clip file on disk -> Feeder(...)[0] (val path: the reference's own Feeder produced the expected sample,
    tests/golden/feeder.npz) -> batch of one -> Model in eval mode -> (1, 10) logits against the oracle fed the
    FIXTURE's sample (so a feeder error and a model error cannot cancel)."""
    i = 7
    name = str(FEEDER_GOLD[f'val/{i}/name'])
    os.makedirs(tmp_path / name)
    with open(tmp_path / name / (name + '.json'), 'w') as f:
        json.dump({'skeletons': FEEDER_GOLD[f'val/{i}/raw'].tolist()}, f)
    fd = Feeder(str(tmp_path), 'val', data_dict=[{'file_name': name, 'label': int(FEEDER_GOLD[f'val/{i}/label']) + 1}])
    data, rgb, label, index = fd[0]
    want = FEEDER_GOLD[f'val/{i}/data']
    assert data.shape == (3, 52, 20, 1) and data.dtype == np.float32 and np.array_equal(data, want)
    assert label == int(FEEDER_GOLD[f'val/{i}/label']) and index == 0
    dev = torch.device('cuda:0')
    m = M.Model(**UCLA)
    fill_state_(m.state_dict(), seed=MODEL_PARAM_SEED)
    sd0 = m.state_dict()
    with torch.no_grad():                                  # realistic running statistics (the ucla_t52 fixture's)
        for k in sd0:
            if 'running_' in k:
                sd0[k].copy_(torch.from_numpy(golden_models[f'ucla_t52/evalbuf/{k}']))
    sd = O.clone_state(m.state_dict())
    ref = O.model_forward(torch.from_numpy(want)[None], sd, 20, training=False)
    m = m.to(dev).eval()
    x = torch.from_numpy(data)[None].to(dev)               # what a DataLoader with batch_size 1 hands the model
    with torch.no_grad():
        logits = m(x)
        feat, _ = m.extract_feature(x)
    assert tuple(logits.shape) == (1, 10) and tuple(feat.shape) == (1, 256, 13, 20, 1)
    err, scale = float((logits.cpu() - ref).abs().max()), max(1.0, float(ref.abs().max()))
    assert err <= 1e-3 * scale, (err, scale, logits.cpu(), ref)
    assert int(logits.argmax(1)) == int(ref.argmax(1))
    # the batch form the data-parallel step uses gives the same clip
    xb, lab, _ = fd.batch([0])
    with torch.no_grad():
        assert torch.equal(m(xb), logits) and lab.tolist() == [label]


# ---------------------------------------------------------------------------------------------------------------------
# config 2
# ---------------------------------------------------------------------------------------------------------------------
def _derive_cpu(x, name):
    """The four streams as the reference's feeder / upstream CTR-GCN define them (feeder_nucla_gcn.py:119-127)."""
    if name == 'joint':
        return x
    pl = torch.tensor(BONE_PARENT, dtype=torch.long)
    bone = x - x[:, :, :, pl, :]
    if name == 'bone':
        return bone
    src = x if name == 'motion' else bone
    out = torch.zeros_like(src)
    out[:, :, :-1] = src[:, :, 1:] - src[:, :, :-1]
    return out


def _four_models(dev):
    models = torch.nn.ModuleList()
    for i in range(4):
        m = M.Model(**UCLA)
        fill_state_(m.state_dict(), seed=MODEL_PARAM_SEED + i)
        models.append(m)
    return models.to(dev).train()


def test_config2_four_streams_one_arena_one_bucket():
    from tam_gcn_amd.distributed import ParamArena, SGDNesterov
    from tam_gcn_amd.functional import CrossEntropyLoss
    dev = torch.device('cuda:0')
    B, T = 6, 64
    x = make_input((B, 3, T, 20, 1), seed=31)
    lab = make_labels(B, 10, seed=32)
    # --- oracle: four independent models on the four derived inputs (CPU, fp64 = what every fp32 evaluation approximates)
    ref_loss, ref_grads, ref_logits = [], [], []
    cpu_models = _four_models(torch.device('cpu'))
    for name, m in zip(STREAMS, cpu_models):
        sd = {k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in m.state_dict().items()}
        for k, _ in m.named_parameters():
            sd[k].requires_grad_(True)
        lo = O.model_forward(_derive_cpu(x, name).double(), sd, 20, training=True)
        loss = torch.nn.functional.cross_entropy(lo, lab)
        loss.backward()
        ref_loss.append(float(loss))
        ref_logits.append(lo.detach())
        ref_grads.append({k: sd[k].grad for k, _ in m.named_parameters()})
    # --- HIP: the step of bench.py --config 4stream
    models = _four_models(dev)
    arena = ParamArena(models)
    bucket = arena.grad_bucket()
    assert arena.total * 4 > 4 * 6.7e6 and arena.intact()          # one 27 MB buffer
    opt = SGDNesterov(arena.params, lr=0.01, momentum=0.9, weight_decay=1e-4, arena=arena, bucket=bucket)
    ce = CrossEntropyLoss()
    xg, lg = x.to(dev), lab.to(dev)
    parent = torch.tensor(BONE_PARENT, dtype=torch.int32, device=dev)
    bucket.zero()
    losses, logits = [], []
    total = None
    for name, m in zip(STREAMS, models):
        xs = xg if name == 'joint' else ops.stream_derive(xg, parent, name)
        assert torch.equal(xs.cpu(), _derive_cpu(x, name))           # the derivation kernel is exact
        out = m(xs)
        loss = ce(out, lg)
        logits.append(out.detach())
        losses.append(loss.detach())
        total = loss if total is None else total + loss
    total.backward()
    flat = bucket.pack()
    torch.cuda.synchronize()
    for i, name in enumerate(STREAMS):
        assert float((logits[i].cpu().double() - ref_logits[i]).abs().max()) <= 1e-3, name
        assert torch.equal(logits[i].argmax(1).cpu(), ref_logits[i].argmax(1)), name
        assert abs(float(losses[i]) - ref_loss[i]) <= 1e-3, name
    # the packed bucket: every parameter's slot holds that parameter's gradient (layout), and each model's segment
    # matches the oracle in the flip-robust metrics of tests/test_gpu_model.py (6 clips: ReLU masks differ between any
    # two fp32 evaluations), fc gradients (forward features only) tightly
    pos = {id(p): (o, p.numel()) for p, o in zip(arena.params, arena.offsets)}
    flat_c = flat.detach().cpu().double()
    for i, (name, m) in enumerate(zip(STREAMS, models)):
        got, ref = [], []
        for k, p in m.named_parameters():
            o, n = pos[id(p)]
            assert p.grad.data_ptr() == flat.data_ptr() + 4 * o                 # .grad is the bucket's view
            got.append(flat_c[o:o + n])
            ref.append(ref_grads[i][k].reshape(-1))
        g, r = torch.cat(got), torch.cat(ref)
        l2 = float((g - r).norm() / r.norm())
        cos = float((g * r).sum() / (g.norm() * r.norm()))
        assert l2 <= 5e-2 and cos >= 0.999, (name, l2, cos)
        assert _rel(m.fc.weight.grad, ref_grads[i]['fc.weight']) <= 1e-3, name
    # padding between aligned runs stays zero (it is all-reduced and fed to the flat optimiser with the rest)
    mask = torch.ones(arena.total, dtype=torch.bool)
    for p, o in zip(arena.params, arena.offsets):
        mask[o:o + p.numel()] = False
    assert float(flat.cpu()[mask].abs().max() if mask.any() else 0.0) == 0.0
    # --- flat SGD on the arena == four torch.optim.SGD on four separate models given the same gradients
    twins = _four_models(dev)
    opts = [torch.optim.SGD(t.parameters(), lr=0.01, momentum=0.9, nesterov=True, weight_decay=1e-4) for t in twins]
    for step in range(2):                                            # two steps: the momentum buffers matter in the second
        if step:
            bucket.zero()
            tot = None
            for name, m in zip(STREAMS, models):
                xs = xg if name == 'joint' else ops.stream_derive(xg, parent, name)
                l_ = ce(m(xs), lg)
                tot = l_ if tot is None else tot + l_
            tot.backward()
            bucket.pack()
        for name, t, o_ in zip(STREAMS, twins, opts):
            o_.zero_grad()
            xs = xg if name == 'joint' else ops.stream_derive(xg, parent, name)
            ce(t(xs), lg).backward()
            o_.step()
        opt.step()
    torch.cuda.synchronize()
    for m, t in zip(models, twins):
        sa, sb = m.state_dict(), t.state_dict()
        assert list(sa.keys()) == list(sb.keys()) and len(sa) == 892
        for k in sa:
            if sa[k].is_floating_point():
                assert _rel(sa[k], sb[k]) <= 2e-5, k
            else:
                assert torch.equal(sa[k], sb[k]), k


def _run_stream_check(mode, n=4, nst=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-X', 'faulthandler', os.path.join(root, 'tools', 'stream_capture_check.py'), mode, str(n), str(nst or n)],
                       capture_output=True, text=True, timeout=600)
    return r.returncode, r.stdout, r.stderr


def test_models_on_separate_streams_eager():
    """Four Models, each on its own HIP stream -- forward and, through autograd, backward -- as
    bench.py --config 4stream runs them: bit-identical to the same step on ONE stream.  Every main stream owns its side
    streams (functional._side_streams): blocks of the caching allocator never move between two models' streams.
    (Own process: tools/stream_capture_check.py.)"""
    rc, out, err = _run_stream_check('eager')
    assert rc == 0 and 'eager: OK' in out, (rc, out[-1500:], err[-3000:])


def _run_stream_check_sized(mode, n, nst, clips, frames):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CHECK_CLIPS=str(clips), CHECK_T=str(frames))
    r = subprocess.run([sys.executable, '-X', 'faulthandler', os.path.join(root, 'tools', 'stream_capture_check.py'), mode, str(n), str(nst)],
                       capture_output=True, text=True, timeout=900, env=env)
    return r.returncode, r.stdout, r.stderr


@pytest.mark.parametrize('n,nst,clips,frames', [(4, 2, 4, 32), (4, 4, 64, 64)], ids=['4on2_small', '4on4_64clips'])
def test_models_on_separate_streams_captured(n, nst, clips, frames):
    """The same step captured into ONE HIP graph and replayed twice: bit-identical to the eager one-stream step, gradients,
    running statistics and losses (the tool fails on ANY difference).  Round 2's form of it (side streams forked from the model
    streams, a two-level fork) faulted inside hipStreamEndCapture; model streams do not fork any more.  At 64 clips x 64 frames
    the four models' kernels really overlap: this is the size at which a register hazard in ctrgc_de_tail (inline-asm loads
    whose registers the compiler copied before they had landed) showed as garbage gradients of one model in one run out of
    two -- and only here, never in a one-stream step."""
    rc, out, err = _run_stream_check_sized('capture', n, nst, clips, frames)
    assert rc == 0 and 'capture: OK' in out and 'DIFFERS' not in out, (rc, out[-1500:], err[-3000:])


def test_bench_runs_model_streams_under_graph_replay():
    """bench.py --config 4stream spreads the four models over forked streams inside its HIP graph (the default there)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--config', '4stream', '--fork-streams', '2', '--batch', '8', '--steps', '2',
                        '--warmup', '1', '--no-cpu-baseline'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr[-1500:])
    line = json.loads([l_ for l_ in r.stdout.splitlines() if l_.startswith('{')][-1])
    assert line['config']['launch'] == 'hipgraph' and line['config'].get('model_streams') == 2, line['config']


def test_capture_on_a_plain_forked_stream_is_guarded():
    """A caller that captures a model under a plain torch.cuda.stream(s) forked from the capture stream (not declared
    through model_stream / allow_side_streams_in_capture): functional._side_ok keeps its blocks from forking during the
    capture -- the step captures and replays bit-identically instead of taking the process down."""
    rc, out, err = _run_stream_check('capture_raw', 2)
    assert rc == 0 and 'capture_raw: OK' in out, (rc, out[-1500:], err[-3000:])


# ---------------------------------------------------------------------------------------------------------------------
# config 4
# ---------------------------------------------------------------------------------------------------------------------
def test_config4_full_size_properties():
    """BASELINE configs[4] at its frame count: ONE TCN_GCN_unit(256, 256) on (32, 256, 512, 64) -- 16 frame chunks of the
    streaming aggregation kernels, the joint-sliced k x 1 convolutions, 1.07 GB per activation.  No oracle run is affordable
    here; the domain's size-independent properties: (1) eval mode: a clip's output does not depend on the batch it sits in
    (32-clip launch vs 2-clip launch); (2) train mode: permuting the batch permutes the output and leaves every parameter
    gradient unchanged; (3) temporal locality: the block's receptive field is +-4 frames, so the eval output of frames
    [0, 200) of a clip is unchanged by what happens in frames >= 204 -- except through CTRGC's mean over T, which is held
    fixed by making the perturbation zero-mean over T per (channel, joint)."""
    from helpers import A_BY_V
    dev = torch.device('cuda:0')
    C, T, V, NB = 256, 512, 64, 32
    torch.manual_seed(0)
    blk = M.TCN_GCN_unit(C, C, A_BY_V[V])
    fill_state_(blk.state_dict(), seed=77)
    blk = blk.to(dev)
    g = torch.Generator().manual_seed(13)
    x = (torch.rand(NB, C, T, V, generator=g) * 2 - 1).to(dev)
    blk.train()
    for mod in blk.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.momentum = 1.0                               # running statistics := this batch's
    with torch.no_grad():
        blk(x)
    blk.eval()
    with torch.no_grad():
        full = blk(x)
        sub = blk(x[10:12].contiguous())
        assert torch.isfinite(full).all()
        scale = float(full.abs().max())
        assert float((full[10:12] - sub).abs().max()) <= 1e-4 * scale
        # (3) a zero-mean-over-T perturbation confined to frames >= 204
        x2 = x[10:12].clone()
        d = (torch.rand(2, C, 8, V, generator=g) * 2 - 1).to(dev)
        x2[:, :, 300:308] += d
        x2[:, :, 400:408] -= d
        y2 = blk(x2)
        assert float((y2[:, :, :200] - sub[:, :, :200]).abs().max()) <= 1e-4 * scale
        assert float((y2[:, :, 300:308] - sub[:, :, 300:308]).abs().max()) > 1e-2 * scale
    del full, y2
    blk.train()
    cot = (torch.rand(NB, C, T, V, generator=g) * 2 - 1).to(dev)
    perm = torch.randperm(NB, generator=g).to(dev)
    xr = x.clone().requires_grad_(True)
    out = blk(xr)
    out.backward(cot)
    g1 = {k: p.grad.clone() for k, p in blk.named_parameters()}
    dx1 = xr.grad
    for p in blk.parameters():
        p.grad = None
    xp = x[perm].contiguous().requires_grad_(True)
    outp = blk(xp)
    outp.backward(cot[perm].contiguous())
    torch.cuda.synchronize()
    assert float((outp.detach() - out.detach()[perm]).abs().max()) <= 2e-4 * float(out.detach().abs().max())
    assert float((xp.grad - dx1[perm]).abs().max()) <= 2e-3 * float(dx1.abs().max())
    gmax = max(float(v.norm()) for v in g1.values())
    for k, p in blk.named_parameters():                     # (biases in front of a train-mode BatchNorm: exact-zero gradients,
        n1 = float(g1[k].norm())                            # rounding noise only -- hence the floor)
        assert float((g1[k] - p.grad).norm()) <= 5e-3 * n1 + 1e-4 * gmax, k


def test_config4_at_its_stated_size_256_clips():
    """BASELINE configs[4] at the per-GPU size it states: ONE TCN_GCN_unit(256, 256) on (256, 256, 512, 64) -- 8.6 GB per
    activation, x3 = conv3(x) has 6.4e9 elements, so the calls are split over clips by ops.n_chunks WITHOUT the test forcing it
    (the un-forced path had never run).  (1) eval mode: clips 100..101 of the 256-clip launch equal the
    same two clips launched alone; (2) train mode: forward + backward finite, and permuting the batch leaves every parameter
    gradient unchanged (fp32 summation-order noise) -- the chunked launches cover every clip exactly once."""
    from helpers import A_BY_V
    from tam_gcn_amd import ops
    dev = torch.device('cuda:0')
    free, total = torch.cuda.mem_get_info()
    if total < 250 * 2 ** 30:
        pytest.skip('needs a 288 GB MI355X')
    C, T, V, NB = 256, 512, 64, 256
    assert NB * 3 * C * T * V > ops.CHUNK_ELEMS          # x3 really exceeds one launch
    torch.manual_seed(0)
    blk = M.TCN_GCN_unit(C, C, A_BY_V[V])
    fill_state_(blk.state_dict(), seed=77)
    blk = blk.to(dev)
    gen = torch.Generator(device=dev).manual_seed(13)
    x = torch.rand(NB, C, T, V, generator=gen, device=dev) * 2 - 1
    blk.eval()
    with torch.no_grad():
        full = blk(x)
        sub = blk(x[100:102].contiguous())
        assert torch.isfinite(full).all()
        scale = float(full.abs().max())
        assert float((full[100:102] - sub).abs().max()) <= 1e-4 * scale
    del full, sub
    blk.train()
    cot = torch.rand(NB, C, T, V, generator=gen, device=dev) * 2 - 1
    xr = x.requires_grad_(True)
    out = blk(xr)
    out.backward(cot)
    assert torch.isfinite(out.detach()).all() and torch.isfinite(xr.grad).all()
    g1 = {k: p.grad.clone() for k, p in blk.named_parameters()}
    del out
    xr.grad = None
    for p in blk.parameters():
        p.grad = None
    perm = torch.randperm(NB, generator=torch.Generator().manual_seed(5)).to(dev)
    xp = x.detach()[perm].contiguous().requires_grad_(True)
    del xr, x
    cp = cot[perm].contiguous()
    del cot
    outp = blk(xp)
    outp.backward(cp)
    torch.cuda.synchronize()
    gmax = max(float(v.norm()) for v in g1.values())
    for k, p in blk.named_parameters():
        assert torch.isfinite(p.grad).all(), k
        assert float((g1[k] - p.grad).norm()) <= 5e-3 * float(g1[k].norm()) + 1e-4 * gmax, k
