"""N>1 path on CPU: world_size-2 gloo processes exercise the flat gradient bucket,
the mean all-reduce, state broadcast and batch sharding."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tam_gcn_amd.distributed import FlatGradBucket, SGDNesterov, broadcast_state, shard_batch


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                       # replicas start different on purpose
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    broadcast_state(net, src=0)
    bucket = FlatGradBucket(net.parameters())
    opt = SGDNesterov(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    g = torch.Generator().manual_seed(7)
    X = torch.randn(8, 6, generator=g)
    Y = torch.randint(0, 3, (8,), generator=g)
    lo, hi = shard_batch(8, rank, world)
    for _ in range(2):
        bucket.zero()
        torch.nn.functional.cross_entropy(net(X[lo:hi]), Y[lo:hi]).backward()
        bucket.pack()
        bucket.all_reduce_mean()
        opt.step()
    out[rank] = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    dist.destroy_process_group()


def test_two_rank_dp_equals_single_process_full_batch():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert torch.allclose(out[0], out[1], atol=0, rtol=0)          # replicas stay identical
    # single-process reference on the full batch with torch's own SGD
    torch.manual_seed(100)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4)
    g = torch.Generator().manual_seed(7)
    X = torch.randn(8, 6, generator=g)
    Y = torch.randint(0, 3, (8,), generator=g)
    for _ in range(2):
        opt.zero_grad()
        torch.nn.functional.cross_entropy(net(X), Y).backward()
        opt.step()
    ref = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    assert torch.allclose(out[0], ref, atol=1e-6, rtol=1e-5)


def test_shard_batch_covers_everything():
    for n, w in [(1024, 8), (10, 4), (3, 8)]:
        spans = [shard_batch(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


def test_param_arena_zero_copy_packing_and_flat_sgd():
    """ParamArena: values preserved, packed operands become views of the arena, state_dict stays per-parameter, and
    the flat optimiser step equals the multi-tensor one."""
    import copy
    from tam_gcn_amd.distributed import ParamArena
    from tam_gcn_amd.models.ctrgcn import Model
    torch.manual_seed(3)
    m = Model(num_class=10, num_point=20, num_person=1, graph='graph.ucla.Graph', graph_args=dict(labeling_mode='spatial'))
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.01 * torch.randn_like(p))
    ref = copy.deepcopy(m)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    arena = ParamArena(m)
    assert arena.intact() and arena.flat.numel() >= sum(p.numel() for p in m.parameters())
    assert all(p.data_ptr() % 16 == 0 for grp in m.l5.gcn1._arena_groups() for p in grp[:1])       # aligned group starts
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k
        assert v.untyped_storage().data_ptr() != arena.flat.untyped_storage().data_ptr(), k      # cloned, not a view
    gcn, tcn = m.l5.gcn1, m.l5.tcn1
    P = gcn._pack(gcn._tensors(torch.device('cpu')))
    assert P.W3.data_ptr() == gcn.convs[0].conv3.weight.data_ptr() and P.W12.data_ptr() == gcn.convs[0].conv1.weight.data_ptr()
    assert torch.equal(P.W3, torch.cat([c.conv3.weight.reshape(gcn.out_c, -1) for c in gcn.convs]))
    assert torch.equal(P.W4, torch.stack([c.conv4.weight.reshape(gcn.out_c, -1) for c in gcn.convs]))
    assert torch.equal(P.B12, torch.cat([b for c in gcn.convs for b in (c.conv1.bias, c.conv2.bias)]))
    Pt = tcn._pack(tcn._tensors())
    assert Pt.Win.data_ptr() == tcn.branches[0][0].weight.data_ptr()
    Pr = ref.l5.gcn1._pack(ref.l5.gcn1._tensors(torch.device('cpu')))                          # no arena: plain cat, same values
    assert torch.equal(Pr.W3, P.W3) and Pr.W3.data_ptr() != ref.l5.gcn1.convs[0].conv3.weight.data_ptr()
    # one optimiser step, flat vs multi-tensor, same gradients
    bucket = arena.grad_bucket()
    opt = SGDNesterov(arena.params, lr=0.1, momentum=0.9, weight_decay=1e-4, arena=arena, bucket=bucket)
    rparams = dict(ref.named_parameters())
    rlist = [rparams[k] for k, _ in m.named_parameters()]
    ropt = SGDNesterov(rlist, lr=0.1, momentum=0.9, weight_decay=1e-4)
    g = torch.Generator().manual_seed(5)
    for _ in range(2):
        for (k, p), rp in zip(m.named_parameters(), rlist):
            gr = torch.randn(p.shape, generator=g)
            p.grad = gr.clone(); rp.grad = gr.clone()
        bucket.pack(); opt.step(); ropt.step()
    for (k, p), rp in zip(m.named_parameters(), rlist):
        assert torch.allclose(p, rp, rtol=1e-6, atol=1e-7), k
    assert arena.intact()
    m.load_state_dict(before)                                    # in-place copies keep the views
    assert arena.intact() and torch.equal(m.fc.weight, before['fc.weight'])


# ---------------------------------------------------------------------------
# The real Model + ParamArena + flat bucket + flat SGD over two gloo ranks with UNEQUAL shards (5 + 3 clips).  The
# HIP forward cannot run here, so each rank's gradients come from the CPU oracle on its shard (test infrastructure);
# everything after the backward is the product's distributed code.
# ---------------------------------------------------------------------------
_MARGS = dict(num_class=10, num_point=20, num_person=1, graph='graph.ucla.Graph', graph_args=dict(labeling_mode='spatial'))


def _oracle_grads(model, x, lab):
    from oracle import ctrgcn_oracle as O
    sd = O.clone_state(model.state_dict(), requires_grad=True)
    loss = torch.nn.functional.cross_entropy(O.model_forward(x, sd, 20, training=True), lab)
    loss.backward()
    return {k: sd[k].grad for k, _ in model.named_parameters()}


def _model_worker(rank, world, port, out):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden'))
    from params import fill_state_, make_input, make_labels
    from tam_gcn_amd.distributed import ParamArena
    from tam_gcn_amd.models.ctrgcn import Model
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    m = Model(**_MARGS)
    fill_state_(m.state_dict(), seed=50 + rank)              # replicas start different on purpose
    arena = ParamArena(m)
    broadcast_state(m, src=0, arena=arena)                    # ONE broadcast of the arena + one per dtype of packed buffers
    out[f's{rank}'] = {k: v.clone() for k, v in m.state_dict().items()}
    bucket = arena.grad_bucket()
    opt = SGDNesterov(arena.params, lr=0.01, momentum=0.9, weight_decay=1e-4, arena=arena, bucket=bucket)
    n_global = 8
    X, Y = make_input((n_global, 3, 8, 20, 1), seed=9), make_labels(n_global, 10, seed=10)
    lo, hi = (0, 5) if rank == 0 else (5, 8)                   # unequal on purpose
    bucket.zero()
    grads = _oracle_grads(m, X[lo:hi], Y[lo:hi])
    for k, p in m.named_parameters():
        p.grad = grads[k].clone()
    bucket.pack()
    bucket.all_reduce_mean(local_n=hi - lo, global_n=n_global)
    out[f'g{rank}'] = bucket.flat.clone()
    opt.step()
    out[rank] = arena.flat.clone()
    assert arena.intact()
    dist.destroy_process_group()


def test_two_rank_real_model_arena_uneven_shards():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden'))
    from params import fill_state_, make_input, make_labels
    from tam_gcn_amd.distributed import ParamArena
    from tam_gcn_amd.models.ctrgcn import Model
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_model_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert torch.equal(out[0], out[1])                          # replicas identical after the step
    assert torch.equal(out['g0'], out['g1'])
    ref = Model(**_MARGS)                                       # the packed broadcast delivered rank 0's whole state: parameters,
    fill_state_(ref.state_dict(), seed=50)                      # running statistics, num_batches_tracked counters
    for r in (0, 1):
        st = out[f's{r}']
        assert list(st.keys()) == list(ref.state_dict().keys())
        assert all(torch.equal(st[k], v) and st[k].dtype == v.dtype for k, v in ref.state_dict().items())
    # single-process emulation: per-shard gradients (per-replica BatchNorm statistics, as nn.DataParallel computes them)
    # weighted by shard size = the gradient of the mean loss over the 8 clips
    m = Model(**_MARGS)
    fill_state_(m.state_dict(), seed=50)
    arena = ParamArena(m)
    bucket = arena.grad_bucket()
    X, Y = make_input((8, 3, 8, 20, 1), seed=9), make_labels(8, 10, seed=10)
    g0, g1 = _oracle_grads(m, X[:5], Y[:5]), _oracle_grads(m, X[5:], Y[5:])
    for k, p in m.named_parameters():
        p.grad = (5 * g0[k] + 3 * g1[k]) / 8
    bucket.pack()
    # (thread counts differ between the workers and this process: fp32 summation order, ~1e-6 of the gradient scale)
    assert torch.allclose(out['g0'], bucket.flat, rtol=1e-4, atol=2e-6 * float(bucket.flat.abs().max()))
    opt = SGDNesterov(arena.params, lr=0.01, momentum=0.9, weight_decay=1e-4, arena=arena, bucket=bucket)
    opt.step()
    assert torch.allclose(out[0], arena.flat, rtol=1e-5, atol=1e-6)


def test_bench_self_launches_two_ranks_rehearsal():
    """`python bench.py --gpus 2` outside a launcher must start its own two ranks and print ONE JSON line with n_gpus 2
    (the driver invokes it exactly like that).  TAMGCN_BENCH_REHEARSAL=1 runs the control flow on CPU tensors over gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['TAMGCN_BENCH_REHEARSAL'] = '1'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout.decode()
    js = json.loads(lines[0])
    assert js['n_gpus'] == 2 and js['steps'] == 3 and js['config']['rehearsal'] is True and js['value'] is None
    assert js['config']['global_batch'] == 512 and js['scaling'] == 'weak'
    assert js['cpu_baseline'] is None and js['cpu_baseline_absent_because']
    # configs[2]: four models, one arena / bucket, one all-reduce -- the same control flow
    r4 = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--config', '4stream'],
                        env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r4.returncode == 0, r4.stderr.decode()[-2000:]
    js4 = json.loads([ln for ln in r4.stdout.decode().splitlines() if ln.startswith('{')][0])
    assert js4['n_gpus'] == 2 and js4['config']['streams'] == 4 and js4['config']['global_batch'] == 256 and js4['value'] is None
    # the segmented, overlapped exchange (distributed.SegmentedReducer) in the same control flow: eager launches, three segments;
    # the stand-in gradients are rank-dependent constants, so the final "loss" and the parameters after the flat SGD steps must
    # equal the one-bucket run's exactly
    ro = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                         '--overlap-allreduce', '3'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert ro.returncode == 0, ro.stderr.decode()[-2000:]
    jso = json.loads([ln for ln in ro.stdout.decode().splitlines() if ln.startswith('{')][0])
    assert jso['n_gpus'] == 2 and jso['config']['allreduce'].startswith('3 segments') and jso['config']['launch'] == 'eager'
    assert js['config']['allreduce'].startswith('one flat bucket')
    ro4 = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--config', '4stream',
                          '--overlap-allreduce', '3'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert ro4.returncode == 0, ro4.stderr.decode()[-2000:]
    # a launcher environment that disagrees with --gpus is an error, not silently ignored
    env2 = dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r2 = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2'], env=env2, stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, timeout=600)
    assert r2.returncode != 0 and b'WORLD_SIZE=1' in r2.stderr


def _segment_worker(rank, world, port, out):
    """The real Model's parameters in a ParamArena; gradients arrive through autograd (a surrogate loss whose gradient w.r.t.
    every parameter is a seeded rank-dependent tensor), so the post-accumulate hooks fire as in a real backward."""
    from tam_gcn_amd.distributed import ParamArena, SegmentedReducer
    from tam_gcn_amd.models.ctrgcn import Model
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)
    models = torch.nn.ModuleList([Model(num_class=10, num_point=20, num_person=1, graph='graph.ucla.Graph',
                                        graph_args=dict(labeling_mode='spatial')) for _ in range(2)])     # two models, one arena
    arena = ParamArena(models)
    bucket = arena.grad_bucket()
    g = torch.Generator().manual_seed(50 + rank)
    coefs = [torch.randn(p.shape, generator=g) for p in arena.params]

    def backward(skip=()):
        bucket.zero()
        loss = sum((p * c).sum() for i, (p, c) in enumerate(zip(arena.params, coefs)) if i not in skip)
        return loss

    # reference: everything packed after backward, ONE all-reduce
    backward().backward()
    bucket.pack()
    ref = bucket.all_reduce_mean().clone()
    # segmented: hooks pack and reduce each segment as its last gradient arrives
    red = SegmentedReducer(bucket, nseg=3)
    assert len(red.ranges) == 3 and red.ranges[0][1] == len(arena.params) and red.ranges[-1][0] == 0
    loss = backward()
    red.begin()
    loss.backward()
    assert all(red.sent)                                   # every segment left during backward
    got = red.finish().clone()
    # parameters without a gradient this step (a frozen branch): their segments leave in finish(), their slots are zero
    skip = {0, 1, len(arena.params) - 1}
    loss = backward(skip)
    red.begin()
    loss.backward()
    assert not all(red.sent)
    got2 = red.finish().clone()
    bucket.zero()
    backward(skip).backward()
    bucket.pack()
    ref2 = bucket.all_reduce_mean().clone()
    red.remove()
    out[rank] = (bool(torch.equal(got, ref)), bool(torch.equal(got2, ref2)), float(ref.abs().sum()),
                 [int(hi - lo) for lo, hi in red.ranges])
    dist.destroy_process_group()


def test_segmented_overlapped_allreduce_equals_the_single_allreduce():
    """the bucket in three contiguous segments (tail first, the order backward completes them), each
    all-reduced when its last gradient has been packed: bit-equal to one all-reduce of the whole bucket, on two gloo ranks,
    with one arena over TWO models (the 4-stream configuration's layout) and with parameters that receive no gradient."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_segment_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        same, same2, mass, sizes = out[r]
        assert same and same2 and mass > 0, out[r]
        assert len(sizes) == 3 and min(sizes) > 0
