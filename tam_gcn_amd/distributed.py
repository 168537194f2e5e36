"""Data-parallel step for the clip-sharded CTR-GCN.

One process per GPU; parameters are replicated; every parameter's ``.grad`` is a
view into ONE flat fp32 bucket (1,693,260 floats = 6.77 MB for the N-UCLA model), so
the whole gradient exchange is a single in-place RCCL all-reduce over xGMI (or gloo
on CPU in the tests) with no packing copies.  BatchNorm statistics stay per replica,
which is what the reference's nn.DataParallel does (processor/io.py:86-87).
"""
import torch
import torch.distributed as dist


class ParamArena:
    """All trainable parameters in ONE flat fp32 buffer; every ``p.data`` becomes a view of it (values preserved).

    Two things follow.  (1) Modules that feed several parameters to one kernel as a concatenated operand (the three
    subsets' conv3 weights, the temporal branches' entry convs, ...: ``_arena_groups()``) find them already back to
    back: their per-forward ``torch.cat`` becomes a zero-copy view (60 launches per step).  (2) With a FlatGradBucket in
    the same order the optimiser is four element-wise kernels on two flat buffers instead of ~40 multi-tensor launches.
    Build it AFTER the model is on its device (``.to()`` re-allocates parameters and would orphan the arena), before
    graph capture.  ``state_dict()`` keeps returning per-parameter tensors (cloned, so a checkpoint does not drag the
    whole arena along); ``load_state_dict`` copies in place as usual."""

    ALIGN = 16                      # floats: every group / stand-alone parameter starts on a 64-byte boundary (the kernels
    #                                 take 16-byte vector and LDS-DMA paths only for 16-byte aligned operands)

    def __init__(self, model):
        seen, order, starts = set(), [], []          # starts[i]: parameter i opens a new aligned run
        for m in model.modules():
            if hasattr(m, '_arena_groups'):
                for grp in m._arena_groups():
                    first = True
                    for p in grp:
                        if p.requires_grad and id(p) not in seen:
                            seen.add(id(p)); order.append(p); starts.append(first)
                            first = False
        for p in model.parameters():
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p)); order.append(p); starts.append(True)
        if not order:
            raise ValueError('no trainable parameters')
        dev, dt = order[0].device, order[0].dtype
        if any(p.device != dev or p.dtype != dt for p in order):
            raise ValueError('ParamArena needs all parameters on one device with one dtype')
        self.params, self.offsets = order, []
        off = 0
        for p, st in zip(order, starts):
            if st:
                off = (off + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            self.offsets.append(off)
            off += p.numel()
        self.total = (off + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.flat = torch.zeros(self.total, device=dev, dtype=dt)
        self.epoch = 0              # see state_version()
        with torch.no_grad():
            for p, o in zip(order, self.offsets):
                v = self.flat[o:o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                p._tamgcn_arena = self          # the eval-mode caches key on state_version(): a write to `flat` does not bump p._version
        ptr = self.flat.untyped_storage().data_ptr()

        def clone_views(module, state, prefix, meta):
            for k, v in list(state.items()):
                if isinstance(v, torch.Tensor) and v.untyped_storage().data_ptr() == ptr:
                    state[k] = v.clone()
        self._hook = model._register_state_dict_hook(clone_views)

    def state_version(self):
        """Changes whenever the parameter VALUES may have changed through the flat buffer.  ``p.data`` are views of ``flat``
        taken under no_grad, so an in-place update of ``flat`` (flat SGD, a broadcast, a load into it) leaves every
        ``p._version`` where it was; caches of folded parameters (f2.FusedEval, functional._eval_cached) add this pair to
        their key.  ``flat._version`` sees eager in-place writes; ``epoch`` is bumped by SGDNesterov.step and
        broadcast_state, and by ``touch()``, which a caller runs after anything Python cannot see: the REPLAY of a HIP
        graph that holds the optimiser step executes no Python, so bump it after ``graph.replay()`` before an eval pass."""
        return (self.flat._version, self.epoch)

    def touch(self):
        self.epoch += 1

    def intact(self):
        """False once something (``model.to``, manual re-assignment) has detached the parameters from the arena."""
        base = self.flat.data_ptr()
        return all(p.data_ptr() == base + o * p.element_size() for p, o in zip(self.params, self.offsets))

    def grad_bucket(self):
        """A FlatGradBucket with this arena's layout (same offsets, padding included): flat SGD needs them congruent."""
        return FlatGradBucket(self.params, offsets=self.offsets, total=self.total)


class FlatGradBucket:
    """All gradients in ONE flat fp32 buffer => one all-reduce, no per-tensor collectives.

    Autograd hands each parameter a freshly produced gradient tensor (``p.grad`` is None before
    backward, so AccumulateGrad steals it: no 600 tiny in-place-add kernels per step); ``pack()``
    gathers them into the bucket with one multi-tensor copy and re-points ``p.grad`` at the
    bucket's views, which is what the optimiser then reads."""

    def __init__(self, params, offsets=None, total=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        dev, dt = self.params[0].device, self.params[0].dtype
        if offsets is None:
            offsets, off = [], 0
            for p in self.params:
                offsets.append(off)
                off += p.numel()
            total = off
        self.offsets = list(offsets)
        self.flat = torch.zeros(total, device=dev, dtype=dt)
        self.views = [self.flat[o:o + p.numel()].view_as(p) for p, o in zip(self.params, self.offsets)]

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

    def all_reduce_mean(self, group=None, local_n=None, global_n=None):
        """Gradient of the mean loss over the GLOBAL batch, as DataParallel's loss over the gathered batch gives it
        (processor/io.py:86-87).  Equal shards (the default): sum over ranks / world size.  Unequal shards
        (shard_batch() with n_global % world != 0): pass this rank's ``local_n`` and the ``global_n``; every rank's
        mean-loss gradient is weighted by local_n / global_n before the sum."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            if local_n is not None:
                if not global_n:
                    raise ValueError('all_reduce_mean: local_n needs global_n')
                self.flat.mul_(float(local_n) / float(global_n))
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
                self.flat.div_(dist.get_world_size(group))
        return self.flat


class SegmentedReducer:
    """The gradient exchange of a FlatGradBucket in `nseg` contiguous segments, each all-reduced on a communication stream
    AS SOON AS backward has produced its last gradient ("2-3 buckets overlapped with backward").

    Backward reaches the blocks l10 -> l1, i.e. the bucket from its END to its start (the arena / bucket order follows
    ``model.modules()``: data_bn, l1, ..., l10, fc), so segment 0 is the TAIL of the flat buffer and completes first.  A
    post-accumulate hook on every parameter counts its segment down; the hook of a segment's last gradient packs that
    segment into the bucket (one multi-tensor copy, ``p.grad`` re-pointed at the bucket's views) and starts its all-reduce
    on the communication stream behind an event; ``finish()`` packs what is left (parameters that received no gradient),
    joins the communication stream and applies the mean.  Element for element the same sums as ONE all-reduce of the whole
    bucket (tests/test_distributed_cpu.py holds the two bit-equal on gloo).

    Eager launches only: under HIP-graph replay no Python (and no hook) runs, the step's two graphs keep ONE all-reduce
    between them (bench.py), which at 6.8 MB is ~0.15 ms of a ~31 ms step."""

    def __init__(self, bucket, nseg=3, group=None, comm_stream=None):
        self.bucket, self.group = bucket, group
        n = len(bucket.params)
        nseg = max(1, min(int(nseg), n))
        # contiguous runs of parameters with about equal element counts, taken from the END of the bucket
        total = sum(p.numel() for p in bucket.params)
        bounds, acc, target = [n], 0, total / nseg
        for i in range(n - 1, 0, -1):
            acc += bucket.params[i].numel()
            if acc >= target * len(bounds) and len(bounds) < nseg:
                bounds.append(i)
        bounds.append(0)
        self.ranges = [(bounds[k + 1], bounds[k]) for k in range(len(bounds) - 1)]       # segment k = params[lo:hi], k = 0 is the tail
        self.seg_of = {}
        for k, (lo, hi) in enumerate(self.ranges):
            for i in range(lo, hi):
                self.seg_of[id(bucket.params[i])] = k
        ends = [bucket.offsets[hi] if hi < n else bucket.flat.numel() for _, hi in self.ranges]
        self.flat_ranges = [(bucket.offsets[lo], e) for (lo, _), e in zip(self.ranges, ends)]
        self.comm = comm_stream
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in bucket.params]
        self.begin()

    def begin(self):
        """Call before every backward (after ``bucket.zero()``)."""
        self.left = [hi - lo for lo, hi in self.ranges]
        self.sent = [False] * len(self.ranges)
        self.scale = None

    def _distributed(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def _pack(self, k):
        lo, hi = self.ranges[k]
        b = self.bucket
        pairs = [(b.views[i], b.params[i].grad) for i in range(lo, hi)]
        f0, f1 = self.flat_ranges[k]
        if any(g is None for _, g in pairs):
            b.flat[f0:f1].zero_()
        live = [(v, g) for v, g in pairs if g is not None and g.data_ptr() != v.data_ptr()]
        if live:
            torch._foreach_copy_([v for v, _ in live], [g for _, g in live])
        for i in range(lo, hi):
            b.params[i].grad = b.views[i]

    def _send(self, k, local_n=None, global_n=None):
        self._pack(k)
        self.sent[k] = True
        if not self._distributed():
            return
        f0, f1 = self.flat_ranges[k]
        seg = self.bucket.flat[f0:f1]
        if self.comm is not None and seg.is_cuda:
            self.comm.wait_stream(torch.cuda.current_stream(seg.device))
            with torch.cuda.stream(self.comm):
                dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group)
        else:
            dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group)

    def _on_grad(self, p):
        k = self.seg_of.get(id(p))
        if k is None or self.sent[k]:
            return
        self.left[k] -= 1
        if self.left[k] == 0:
            self._send(k)

    def finish(self):
        """Segments whose hooks did not all fire (unused parameters) go now; join the communication stream; mean."""
        for k in range(len(self.ranges)):
            if not self.sent[k]:
                self._send(k)
        flat = self.bucket.flat
        if self.comm is not None and flat.is_cuda:
            torch.cuda.current_stream(flat.device).wait_stream(self.comm)
        if self._distributed():
            flat.div_(dist.get_world_size(self.group))
        return flat

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def broadcast_state(module, src=0, group=None, arena=None):
    """Make every replica start from rank ``src``'s parameters and buffers.

    A handful of collectives instead of one per tensor (892 for the N-UCLA model): with ``arena`` (a ParamArena over
    this module) its flat buffer goes in ONE broadcast; everything else -- frozen parameters, BatchNorm running
    statistics, ``num_batches_tracked`` counters -- is packed per dtype into one flat tensor, broadcast, and copied back
    in place (two more broadcasts: fp32 and int64)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    rest = list(module.parameters()) + list(module.buffers())
    if arena is not None:
        if not arena.intact():
            raise ValueError('broadcast_state: the ParamArena no longer backs the parameters')
        dist.broadcast(arena.flat, src=src, group=group)
        arena.touch()
        inside = {id(p) for p in arena.params}
        rest = [t for t in rest if id(t) not in inside]
    by_type = {}
    for t in rest:
        by_type.setdefault((t.dtype, t.device), []).append(t.data)
    with torch.no_grad():
        for ts in by_type.values():
            flat = torch.cat([t.reshape(-1) for t in ts])
            dist.broadcast(flat, src=src, group=group)
            off = 0
            for t in ts:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()


def shard_batch(n_global, rank, world):
    """Contiguous equal split of the clip batch; the remainder goes to the first ranks."""
    base, rem = divmod(n_global, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class SGDNesterov:
    """The reference's optimiser recipe (processor/recognition_rgb.py:23-28: SGD, momentum 0.9,
    nesterov, weight decay) as capture-safe multi-tensor updates on the bucket's views."""

    def __init__(self, params, lr=0.1, momentum=0.9, weight_decay=1e-4, arena=None, bucket=None):
        """With ``arena`` and a ``bucket`` built over ``arena.params`` the update runs on the two flat buffers."""
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.momentum, self.wd = lr, momentum, weight_decay
        self.arena, self.bucket = None, None
        if arena is not None and bucket is not None:
            if [id(p) for p in arena.params] != [id(p) for p in bucket.params] or list(arena.offsets) != list(bucket.offsets):
                raise ValueError('SGDNesterov: build the bucket with arena.grad_bucket() (same order and offsets)')
            if not arena.intact():
                raise ValueError('SGDNesterov: the ParamArena no longer backs the parameters')
            self.arena, self.bucket = arena, bucket
            self.flat_buf = torch.zeros_like(arena.flat)
        else:
            self.bufs = [torch.zeros_like(p) for p in self.params]

    @torch.no_grad()
    def step(self):
        if self.arena is not None:
            p, g, buf = self.arena.flat, self.bucket.flat, self.flat_buf
            self.arena.touch()
            d = torch.add(g, p, alpha=self.wd)                          # g + wd * p
            buf.mul_(self.momentum).add_(d)                             # buf = m*buf + d
            d.add_(buf, alpha=self.momentum)                            # d + m*buf (nesterov)
            p.add_(d, alpha=-self.lr)
            return
        grads = [p.grad for p in self.params]
        d = torch._foreach_add(grads, self.params, alpha=self.wd)      # g + wd * p
        torch._foreach_mul_(self.bufs, self.momentum)
        torch._foreach_add_(self.bufs, d)                               # buf = m*buf + d
        torch._foreach_add_(d, self.bufs, alpha=self.momentum)          # d + m*buf (nesterov)
        torch._foreach_add_(self.params, d, alpha=-self.lr)
