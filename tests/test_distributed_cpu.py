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
