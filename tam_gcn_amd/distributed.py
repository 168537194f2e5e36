"""Data-parallel step for the clip-sharded CTR-GCN.

One process per GPU; parameters are replicated; every parameter's ``.grad`` is a
view into ONE flat fp32 bucket (1,693,260 floats = 6.77 MB for the N-UCLA model), so
the whole gradient exchange is a single in-place RCCL all-reduce over xGMI (or gloo
on CPU in the tests) with no packing copies.  BatchNorm statistics stay per replica,
which is what the reference's nn.DataParallel does (processor/io.py:86-87).
"""
import torch
import torch.distributed as dist


class FlatGradBucket:
    """All gradients in ONE flat fp32 buffer => one all-reduce, no per-tensor collectives.

    Autograd hands each parameter a freshly produced gradient tensor (``p.grad`` is None before
    backward, so AccumulateGrad steals it: no 600 tiny in-place-add kernels per step); ``pack()``
    gathers them into the bucket with one multi-tensor copy and re-points ``p.grad`` at the
    bucket's views, which is what the optimiser then reads."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        dev, dt = self.params[0].device, self.params[0].dtype
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, device=dev, dtype=dt)
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def zero(self):
        for p in self.params:
            p.grad = None

    def pack(self):
        grads = [p.grad for p in self.params]
        if any(g is None for g in grads):                  # parameter unused this step
            self.flat.zero_()
            pairs = [(v, g) for v, g in zip(self.views, grads) if g is not None]
            if pairs:
                torch._foreach_copy_([v for v, _ in pairs], [g for _, g in pairs])
        else:
            torch._foreach_copy_(self.views, grads)
        for p, v in zip(self.params, self.views):
            p.grad = v
        return self.flat

    def all_reduce_mean(self, group=None):
        """Sum over ranks / world size: the gradient of the mean loss over the global batch
        (equal shards), matching DataParallel's loss over the gathered batch."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(dist.get_world_size(group))
        return self.flat


def broadcast_state(module, src=0, group=None):
    """Make every replica start from rank ``src``'s parameters and buffers."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def shard_batch(n_global, rank, world):
    """Contiguous equal split of the clip batch; the remainder goes to the first ranks."""
    base, rem = divmod(n_global, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class SGDNesterov:
    """The reference's optimiser recipe (processor/recognition_rgb.py:23-28: SGD, momentum 0.9,
    nesterov, weight decay) as capture-safe multi-tensor updates on the bucket's views."""

    def __init__(self, params, lr=0.1, momentum=0.9, weight_decay=1e-4):
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.momentum, self.wd = lr, momentum, weight_decay
        self.bufs = [torch.zeros_like(p) for p in self.params]

    @torch.no_grad()
    def step(self):
        grads = [p.grad for p in self.params]
        d = torch._foreach_add(grads, self.params, alpha=self.wd)      # g + wd * p
        torch._foreach_mul_(self.bufs, self.momentum)
        torch._foreach_add_(self.bufs, d)                               # buf = m*buf + d
        torch._foreach_add_(d, self.bufs, alpha=self.momentum)          # d + m*buf (nesterov)
        torch._foreach_add_(self.params, d, alpha=-self.lr)
