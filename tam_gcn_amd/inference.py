"""HIP-graph replay of the eval-mode forward for the inference-only callers (the ensemble evaluation
loop ensemble/ensemble_ctrgcn_resnet_eval.py:147-183, the frozen backbone of models/resnet_gcn_attention.py:82-85,
visual.py:53-55).  Those loops call model(data) with one batch shape over and over; at small batches the launch-fused
eval path is bound by its ~118 launches, and a captured graph replays them without the host in between (batch 1: 4.3 ms
eager -> 2.8 ms, tools/infer_bench.py).

    fast = GraphedForward(model)            # model.eval(), parameters frozen for the lifetime of the capture
    for data, ... in loader:
        logits = fast(data.float().cuda())  # first call per input shape captures, later calls replay

The output tensor is the graph's static buffer: it is overwritten by the next call with the same shape (clone it to keep
it).  Parameter VALUES may change between calls (the graph reads them through their pointers; the eval path's folded
BatchNorm coefficients are keyed on parameter versions, so re-capture with .reset() after loading a new state dict)."""
import torch

__all__ = ['GraphedForward']


class GraphedForward:
    """split (default 1): eval-mode BatchNorm uses the running statistics, so a batch may be cut into `split` slices that run
    side by side on their own HIP streams inside the graph (functional.model_stream) and land in one output tensor -- kernels that
    wait for operands leave room that another slice's kernels fill (four models side by side in one graph run 23 % faster than
    one after the other: DESIGN.md §3 lessons).  Measured on the N-UCLA model: 256 clips 9.36 -> 8.74 ms, 512 clips 17.8 -> 16.5 ms
    with split = 4.  Slices are kept at >= 64 clips (smaller ones would take the latency-oriented f2 kernels, which lose on
    throughput); only tensor-valued methods (forward)."""

    def __init__(self, model, method='forward', max_shapes=8, split=1):
        if model.training:
            raise ValueError('GraphedForward: put the model in eval() mode first (train mode updates running statistics)')
        self.model, self.method, self.max_shapes, self.split = model, method, max_shapes, int(split)
        self._graphs = {}
        self._streams = None

    def reset(self):
        self._graphs.clear()

    def _capture(self, x):
        fn = getattr(self.model, self.method)
        static_in = x.clone()
        with torch.no_grad():
            s = torch.cuda.Stream(device=x.device)
            s.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(s):                       # warm-up off the capture: allocator, side streams, cached coefficients
                for _ in range(2):
                    fn(static_in)
            torch.cuda.current_stream(x.device).wait_stream(s)
            g = torch.cuda.CUDAGraph()
            nsl = max(1, min(self.split, x.shape[0] // 64))
            if nsl > 1:
                from . import functional as Fn
                if self._streams is None:
                    self._streams = [torch.cuda.Stream(device=x.device) for _ in range(self.split)]
                parts = static_in.chunk(nsl)
                with torch.cuda.graph(g):
                    cur = torch.cuda.current_stream(x.device)
                    outs = []
                    for st, xp in zip(self._streams, parts):
                        st.wait_stream(cur)
                        with Fn.model_stream(st):
                            outs.append(fn(xp))
                    for st in self._streams[:len(parts)]:
                        cur.wait_stream(st)
                    out = torch.cat(outs)
            else:
                with torch.cuda.graph(g):
                    out = fn(static_in)
        # The graph reads the eval path's folded BatchNorm coefficients through their pointers, and those tensors are
        # owned by the per-module caches (functional._eval_cached): an eager forward after a parameter change evicts
        # them.  This entry keeps them alive, so a replay without reset() reads stale coefficients, never freed memory.
        keep = [dict(m.__dict__['_tamgcn_eval_cache']) for m in self.model.modules() if '_tamgcn_eval_cache' in m.__dict__]
        eng = self.model.__dict__.get('_tamgcn_f2')        # small batches: the folded weights of tam_gcn_amd.f2, likewise
        if eng:
            keep.append(eng._blocks)
        return g, static_in, out, keep

    def __call__(self, x):
        if not x.is_cuda:
            raise RuntimeError('GraphedForward: expected a HIP (cuda) tensor; there is no CPU path')
        key = (tuple(x.shape), x.dtype, x.device.index)
        ent = self._graphs.get(key)
        if ent is None:
            if len(self._graphs) >= self.max_shapes:
                self._graphs.pop(next(iter(self._graphs)))
            ent = self._graphs[key] = self._capture(x)
        g, static_in, out, _ = ent
        static_in.copy_(x)
        g.replay()
        return out
