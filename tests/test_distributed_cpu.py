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
