"""GPU-box tool and test helper: several Models, each on its own HIP stream (forward and, through autograd, backward),
eagerly and captured into ONE HIP graph -- the step of `bench.py --config 4stream`.  Runs in its own process (a fault inside
hipStreamEndCapture would otherwise take the test session down with it) and prints one line per stage.

    python -X faulthandler tools/stream_capture_check.py <mode> [n_models] [n_streams]     (models i, i + n_streams, ... share a stream)

modes   eager        n models on n streams, eager; must equal the one-stream step bit for bit
        capture      the same step captured (torch.cuda.graph) and replayed twice; bit-identical to eager
        capture_fwd  forward + loss only under capture (no autograd thread)
        capture_noside   capture with the per-block side streams off (TAMGCN_SIDE_STREAMS=0 semantics)
        capture_raw  capture with plain torch.cuda.stream(s) instead of functional.model_stream(s): the guard in
                     functional._side_ok must then keep the model streams from forking (two-level forks fault in HIP)
        capture_twolevel   the faulting form itself: side streams forced on under the model streams (diagnosis only)
        capture_one  ONE model on a stream forked from the capture stream
        capture_relaxed  capture with capture_error_mode='relaxed'
        toy          two torch.nn.Linear stacks on two streams, forward + backward captured (no tam_gcn_amd kernels)
        capture_perbwd   every model's backward is its own .backward() call INSIDE that model's stream context (root gradient created
                     there, no cross-stream hand-over by the autograd engine) instead of one backward of the summed loss
        capture_keepalive   capture while EVERY tensor any operator (or tam_gcn_amd.ops.empty) produces is kept alive until the
                     capture ends: no block of the capture's memory pool is ever handed out twice.  If the replay is then
                     bit-identical, what the plain capture suffers from is a block reused across streams without an edge in
                     the graph (a lifetime problem), not an ordering problem of the kernels themselves
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import torch                                                             # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'capture'
nm = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nst = int(sys.argv[3]) if len(sys.argv) > 3 else nm
dev = torch.device('cuda:0')


def say(msg):
    print(msg, flush=True)


def toy():
    torch.manual_seed(0)
    nets = [torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.ReLU(), torch.nn.Linear(64, 8)).to(dev) for _ in range(2)]
    x = torch.randn(16, 64, device=dev)
    sts = [torch.cuda.Stream(dev) for _ in nets]

    def step():
        cur = torch.cuda.current_stream()
        for n_ in nets:
            for p in n_.parameters():
                p.grad = None
        losses = []
        for n_, st in zip(nets, sts):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                losses.append(n_(x).square().mean())
        for st in sts:
            cur.wait_stream(st)
        (losses[0] + losses[1]).backward()
        return [l_.detach() for l_ in losses]

    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            ref = step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    say('toy: eager ok')
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    say('toy: captured')
    g.replay()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(out, ref))
    say('toy: replay ok')


def main():
    if mode == 'toy':
        return toy()
    from cases import MODEL_CASES, MODEL_PARAM_SEED
    from params import fill_state_, make_input, make_labels
    from tam_gcn_amd import functional as Fn, ops
    from tam_gcn_amd.functional import CrossEntropyLoss
    from tam_gcn_amd.feeder.feeder_nucla_gcn import BONE_PARENT
    from tam_gcn_amd.models import ctrgcn as M
    if mode == 'capture_noside':
        Fn.USE_SIDE_STREAMS = False
    if mode == 'capture_twolevel':
        Fn._side_ok = lambda device, main: True
    on_stream = torch.cuda.stream if mode in ('capture_raw', 'capture_twolevel', 'capture_noside') else Fn.model_stream
    names = ('joint', 'bone', 'motion', 'bone_motion')[:nm]
    models = torch.nn.ModuleList()
    for i in range(nm):
        m = M.Model(**MODEL_CASES[0][1])
        fill_state_(m.state_dict(), seed=MODEL_PARAM_SEED + i)
        models.append(m)
    models = models.to(dev).train()
    NB, NT = int(os.environ.get('CHECK_CLIPS', '4')), int(os.environ.get('CHECK_T', '32'))       # the test's size; bench.py's is 128 x 64
    x = make_input((NB, 3, NT, 20, 1), seed=41).to(dev)
    lab = make_labels(NB, 10, seed=42).to(dev)
    parent = torch.tensor(BONE_PARENT, dtype=torch.int32, device=dev)
    xs = [x if n == 'joint' else ops.stream_derive(x, parent, n) for n in names]
    state0 = [{k: v.clone() for k, v in m.state_dict().items()} for m in models]
    ce = CrossEntropyLoss()
    backward = mode != 'capture_fwd'

    def reset():
        for m, s in zip(models, state0):
            m.load_state_dict(s)

    def step(streams):
        cur = torch.cuda.current_stream()
        for m in models:
            for p in m.parameters():
                p.grad = None
        losses, seen = [], []
        for m, xi, st in zip(models, xs, streams):
            if st is None:
                losses.append(ce(m(xi), lab))
                continue
            if st not in seen:
                st.wait_stream(cur)
                seen.append(st)
            with on_stream(st):
                losses.append(ce(m(xi), lab))
        if mode == 'capture_perbwd':
            for m, l_, st in zip(models, losses, streams):
                if st is None:
                    l_.backward()
                else:
                    with on_stream(st):
                        l_.backward()
            for st in seen:
                cur.wait_stream(st)
            return [l_.detach() for l_ in losses]
        for st in seen:
            cur.wait_stream(st)
        if backward:
            total = losses[0]
            for l_ in losses[1:]:
                total = total + l_
            total.backward()                     # autograd replays every node on its forward's stream and joins the leaf streams
        return [l_.detach() for l_ in losses]

    def snapshot():
        return ([[None if p.grad is None else p.grad.clone() for p in m.parameters()] for m in models],
                [{k: v.clone() for k, v in m.state_dict().items()} for m in models])

    pnames = [[k for k, _ in m.named_parameters()] for m in models]

    def same(a, b, verbose=True):
        (ga, sa), (gb, sb) = a, b
        bad = []
        for i, (x_, y_) in enumerate(zip(ga, gb)):
            for k, p, q in zip(pnames[i], x_, y_):
                if (p is None) != (q is None) or (p is not None and not torch.equal(p, q)):
                    d = float('nan') if p is None or q is None else float((p - q).abs().max() / (q.abs().max() + 1e-30))
                    bad.append((f'model {i} grad {k}', d))
        for i, (s, t) in enumerate(zip(sa, sb)):
            for k, v in s.items():
                if not torch.equal(v, t[k]):
                    bad.append((f'model {i} state {k}', float((v.double() - t[k].double()).abs().max() / (t[k].double().abs().max() + 1e-30))))
        if bad and verbose:
            say(f'   {len(bad)} tensors differ, worst rel {max(d for _, d in bad):.2e}; first: ' + '; '.join(f'{n} (rel {d:.2e})' for n, d in bad[:4]))
            say('   per model: ' + ', '.join(f'{i}: {sum(1 for n, _ in bad if n.startswith(f"model {i} "))}' for i in range(nm)))
            kinds = {}
            for n_, d in bad:
                w = n_.split()
                key = w[2] + ' ' + ('.'.join(w[3].split('.')[:1]) if w[2] == 'grad' else w[3].split('.')[-1])
                kinds.setdefault(key, []).append(d)
            say('   by kind: ' + '; '.join(f'{k}: {len(v)} (worst {max(v):.1e})' for k, v in sorted(kinds.items())))
            top = [n_ for n_, _ in bad if ' grad l10.' in n_ or ' grad fc' in n_ or ' state l10.' in n_ or ' state data_bn' in n_][:24]
            say('   top of the network: ' + '; '.join(f'{n_} ({d:.1e})' for n_, d in bad if n_ in top))
        same.worst = max([d for _, d in bad], default=0.0)
        return not bad

    ref_loss = step([None] * nm)
    torch.cuda.synchronize()
    ref = snapshot()
    say(f'{mode}: one-stream reference done ({nm} models, {nst} streams)')
    pool = [torch.cuda.Stream(dev) for _ in range(nst)]
    sts = [pool[i % nst] for i in range(nm)]
    reset()
    loss = step(sts)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(loss, ref_loss)) and same(snapshot(), ref), 'eager multi-stream step differs'
    keys = [k for k in Fn._SIDE if k[0] == dev.index]
    handles = [s.cuda_stream for k in keys for s in Fn._SIDE[k]]
    assert len(handles) == len(set(handles)), 'a side stream serves two main streams'
    say(f'{mode}: eager multi-stream step bit-identical; {len(keys)} side pools, {len(handles)} side streams')
    if mode == 'eager':
        return
    reset()
    g = torch.cuda.CUDAGraph()
    kw = dict(capture_error_mode='relaxed') if mode == 'capture_relaxed' else {}
    import contextlib
    keep, ctx = [], contextlib.nullcontext()
    if mode == 'capture_keepalive':
        from torch.utils._python_dispatch import TorchDispatchMode
        from torch.utils._pytree import tree_flatten

        class Keep(TorchDispatchMode):
            def __torch_dispatch__(self, func, types, args=(), kwargs=None):
                out = func(*args, **(kwargs or {}))
                keep.extend(t for t in tree_flatten(out)[0] if isinstance(t, torch.Tensor))
                return out
        ctx = Keep()
        real_bwd = torch.Tensor.backward
    with torch.cuda.graph(g, **kw):
        with ctx:
            closs = step(sts)
    if keep:
        say(f'{mode}: {len(keep)} tensors ({sum(t.numel() * t.element_size() for t in keep) / 2 ** 20:.0f} MiB) held until the capture ended')
    say(f'{mode}: captured')
    cgrads = [[p.grad for p in m.parameters()] for m in models]
    for it in range(2):
        reset()
        g.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(closs, ref_loss)), 'captured losses differ'
        exact = True
        if not backward:
            got = ([[None] * len(pn) for pn in pnames], [{k: v for k, v in m.state_dict().items()} for m in models])
            exact = same(got, ([[None] * len(pn) for pn in pnames], ref[1]))
        if backward:
            got = (cgrads, [{k: v for k, v in m.state_dict().items()} for m in models])
            exact = same(got, ref)
            assert exact, 'captured step differs from the eager step'
        say(f'{mode}: replay {it} bit-identical')


if __name__ == '__main__':
    main()
    say(f'{mode}: OK')
