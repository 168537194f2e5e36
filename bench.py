#!/usr/bin/env python
"""bench.py — skeleton clips/sec (fwd + CE loss + bwd [+ gradient all-reduce] + SGD step)
of models.ctrgcn.Model on the N-UCLA joint stream (20 joints x 64 frames), BASELINE.json
configs[1]: batch 256 per GPU, synthetic U(-1,1) clips, seeded init (de-degenerated so
every branch of the block does real work).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One rank per GPU; the clip batch is sharded (weak scaling: 256 clips per GPU), gradients
are averaged with ONE flat-bucket RCCL all-reduce (6.77 MB) per step.  The step is
captured into a HIP graph (fwd+bwd, then the optimiser) and replayed; rank 0 prints one
JSON line.  Besides the throughput it reports
  roofline     — the kernel with the largest share of the step, timed live with HIP events
                 around each of its launches during instrumented (eager) steps;
  cpu_baseline — the CPU oracle (stock-PyTorch restatement of the reference) on the box's
                 host cores, bounded to ~20 s.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch                                            # noqa: E402
import torch.distributed as dist                        # noqa: E402

HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md chip table
F32_MFMA_PEAK = 157.3e12   # FLOP/s, v_mfma_f32_16x16x4_f32 (= fp32 vector peak)

MODEL_ARGS = dict(num_class=10, num_point=20, num_person=1, graph='graph.ucla.Graph',
                  graph_args=dict(labeling_mode='spatial'))
T_FRAMES, V_JOINTS, PER_GPU_BATCH = 64, 20, 256


def dedegenerate_(model, seed=0):
    """Reference default init has alpha=0, unit_gcn.bn.weight=1e-6 and a zero offset_conv
    (models/ctrgcn.py:229,240-244): the refinement and offset branches would multiply zeros.
    Give them trained-like magnitudes so the benchmark measures the real arithmetic."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            parts = name.split('.')
            if parts[-1] == 'alpha':
                p.fill_(0.5)
            elif parts[-3:] == ['gcn1', 'bn', 'weight']:
                p.fill_(1.0)
            elif parts[-3:] == ['offset_conv', '0', 'weight']:
                p.copy_(torch.randn(p.shape, generator=g) * (2.0 / p.shape[0]) ** 0.5)


# ---------------------------------------------------------------------------
# per-launch instrumentation of the C ABI (HIP events on the launch stream)
# ---------------------------------------------------------------------------
class _Probe:
    """Proxy around the ctypes library: records an event pair around every ABI launch."""

    def __init__(self, lib):
        self._lib, self.records, self.on = lib, [], False

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith('tamgcn_') or name in ('tamgcn_last_error', 'tamgcn_last_kernel', 'tamgcn_version', 'tamgcn_conv_nparts',
                                                     'tamgcn_ew_nparts', 'tamgcn_ctrgc_lds_bytes'):
            return fn

        def wrapped(*args):
            if not self.on:
                return fn(*args)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            sym = self._lib.tamgcn_last_kernel().decode()
            self.records.append((sym or name, e0, e1, _algorithmic(name, args)))
            return rc
        return wrapped


def _src_reads(s):
    return 1 + (1 if s.x2 else 0)


def _algorithmic(name, args):
    """(bytes, flops) one launch must move / compute, from the descriptor alone
    (SURVEY.md §8d definitions; weights and per-channel vectors ignored)."""
    try:
        d = args[0]._obj
    except AttributeError:
        return (0.0, 0.0)
    if name == 'tamgcn_conv':
        inb = d.N * d.K * d.T_in * d.V * _src_reads(d.src) / max(1, d.up)
        extra = (1 if d.add1 else 0) + (1 if d.add2 else 0) + (1 if d.mask else 0) + (1 if d.aux else 0)
        outb = d.N * d.M * d.T_out * d.V * (1 + extra)
        return (4.0 * (inb + outb), 2.0 * d.N * d.M * d.K * d.KT * d.T_out * d.V / max(1, d.up))
    if name == 'tamgcn_wgrad':
        b = d.N * d.M * d.T_out * d.V * _src_reads(d.gy) + d.N * d.K * d.T_in * d.V * _src_reads(d.src)
        return (4.0 * b, 2.0 * d.N * d.M * d.K * d.KT * d.T_out * d.V)
    if name == 'tamgcn_ctrgc_fwd':
        b = d.N * d.T * d.V * (d.Cin + d.Cout)
        f = d.N * d.S * (2.0 * d.Cin * d.Cout * d.T * d.V + 2.0 * d.R * d.Cout * d.V * d.V + 2.0 * d.Cout * d.T * d.V * d.V)
        return (4.0 * b, f)
    if name == 'tamgcn_ctrgc_bwd_dx3':
        b = d.N * d.T * d.V * (2 * d.Cout + d.S * d.Cout)
        f = d.N * d.S * (2.0 * d.R * d.Cout * d.V * d.V + 2.0 * d.Cout * d.T * d.V * d.V)
        return (4.0 * b, f)
    if name == 'tamgcn_ctrgc_bwd_de':
        b = d.N * d.T * d.V * (d.Cin + 2 * d.Cout)
        f = d.N * d.S * (2.0 * d.Cin * d.Cout * d.T * d.V + 2.0 * d.Cout * d.T * d.V * d.V + 4.0 * d.R * d.Cout * d.V * d.V)
        return (4.0 * b, f)
    return (0.0, 0.0)


def instrumented_pass(step_fn, probe, steps):
    probe.records, probe.on = [], True
    for _ in range(steps):
        step_fn()
    torch.cuda.synchronize()
    probe.on = False
    agg = {}
    for name, e0, e1, (b, f) in probe.records:
        a = agg.setdefault(name, dict(calls=0, ms=0.0, bytes=0.0, flops=0.0))
        a['calls'] += 1
        a['ms'] += e0.elapsed_time(e1)
        a['bytes'] += b
        a['flops'] += f
    return agg


def roofline_of(agg):
    """Roofline object for the kernel symbol with the largest share of step time (symbols as rocprofv3
    prints them, minus the anonymous-namespace prefix and the argument list)."""
    timed = {k: v for k, v in agg.items() if v['bytes'] > 0}
    if not timed:
        return None, {}
    name = max(timed, key=lambda k: timed[k]['ms'])
    a = timed[name]
    sec = a['ms'] * 1e-3
    bw, fl = a['bytes'] / sec, a['flops'] / sec
    t_hbm, t_mfma = a['bytes'] / HBM_PEAK, a['flops'] / F32_MFMA_PEAK
    traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'traffic.json')        # filled from rocprofv3 --pmc passes
    if os.path.exists(tpath):
        t = json.load(open(tpath)).get(name)
        traffic = t.get('hbm_bytes_per_launch') if isinstance(t, dict) else t
    common = dict(kernel=name, launches=a['calls'], avg_launch_us=1e3 * a['ms'] / a['calls'],
                  algorithmic_bytes_per_launch=a['bytes'] / a['calls'], algorithmic_flops_per_launch=a['flops'] / a['calls'],
                  hbm_frac=bw / HBM_PEAK, mfma_f32_frac=fl / F32_MFMA_PEAK, traffic=traffic)
    if t_mfma > t_hbm:
        r = dict(bound='mfma', achieved=fl / 1e12, peak=F32_MFMA_PEAK / 1e12, unit='TFLOP/s', frac=fl / F32_MFMA_PEAK)
    else:
        r = dict(bound='hbm', achieved=bw / 1e9, peak=HBM_PEAK / 1e9, unit='GB/s', frac=bw / HBM_PEAK)
    r.update(common)
    shares = {k: round(v['ms'], 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1]['ms'])}
    return r, shares


# ---------------------------------------------------------------------------
def host_cores():
    """CPU threads this process may really use: cgroup quota, else affinity; the GPU boxes expose
    every logical CPU of the host (256) but grant a 16-CPU share per GPU, and oversubscribing
    OpenMP by 16x makes the baseline crawl, so an unbounded answer is capped at 16."""
    env = os.environ.get('TAMGCN_CPU_THREADS')
    if env:
        return max(1, int(env))
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            n = min(n, max(1, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    return n if n <= 32 else 16


def _log(msg):
    print(f'[bench +{time.perf_counter() - _T0:7.1f}s] {msg}', file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def cpu_baseline(budget_s=15.0):
    """The oracle (kind 'port': our stock-PyTorch restatement, pinned to the reference by
    tests/golden) on the host cores: fwd + CE + bwd + SGD step, B=16 (the reference's batch
    size, config/nucla/gcn.yaml:37), T=64, V=20."""
    from oracle import ctrgcn_oracle as O
    from tam_gcn_amd.models.ctrgcn import Model
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m = Model(**MODEL_ARGS)
    dedegenerate_(m)
    sd = O.clone_state(m.state_dict(), requires_grad=True)
    params = [v for v in sd.values() if v.requires_grad]
    B = 16
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(B, 3, T_FRAMES, V_JOINTS, 1, generator=g) * 2 - 1
    lab = torch.randint(0, 10, (B,), generator=g)

    def step():
        for p in params:
            p.grad = None
        loss = torch.nn.functional.cross_entropy(O.model_forward(x, sd, 20, training=True), lab)
        loss.backward()
        with torch.no_grad():
            for p in params:
                p.add_(p.grad, alpha=-1e-3)

    t0 = time.perf_counter(); step(); t1 = time.perf_counter() - t0
    _log(f'cpu baseline: {cores} threads, first step {t1:.2f}s')
    n = max(1, min(48, int(budget_s / max(t1, 1e-3)) - 1))      # about 10-15 s of CPU work
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    dt = time.perf_counter() - t0
    return dict(value=B * n / dt, unit='clips/s', cores=torch.get_num_threads(), kind='port',
                sample=f'{n} steps of batch {B} (T=64,V=20) fwd+CE+bwd+SGD after 1 warm-up, {dt:.1f}s')


# ---------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=PER_GPU_BATCH, help='clips per GPU')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node N for --gpus N')
    # Rehearsal hooks (a one-GPU box cannot run RCCL between two ranks): TAMGCN_BENCH_ONE_DEVICE=1 puts every rank on
    # cuda:0 and TAMGCN_DIST_BACKEND=gloo moves the collectives to gloo -- same control flow, not a measurement.
    if os.environ.get('TAMGCN_BENCH_ONE_DEVICE') == '1':
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        backend = os.environ.get('TAMGCN_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from tam_gcn_amd import _lib
    from tam_gcn_amd.distributed import FlatGradBucket, ParamArena, SGDNesterov, broadcast_state
    from tam_gcn_amd.models.ctrgcn import Model
    probe = _Probe(_lib.load())
    _lib._lib = probe                                     # every ABI launch goes through the probe

    torch.manual_seed(0)
    model = Model(**MODEL_ARGS)
    dedegenerate_(model)
    model = model.to(dev).train()
    arena = ParamArena(model)                             # parameters in one flat buffer: zero-copy operand packing, flat SGD
    broadcast_state(model)
    bucket = arena.grad_bucket()
    opt = SGDNesterov(arena.params, lr=0.01, momentum=0.9, weight_decay=1e-4, arena=arena, bucket=bucket)
    g = torch.Generator().manual_seed(1234 + rank)
    B = args.batch
    x = (torch.rand(B, 3, T_FRAMES, V_JOINTS, 1, generator=g) * 2 - 1).to(dev)
    lab = torch.randint(0, 10, (B,), generator=g).to(dev)
    loss_buf = torch.zeros((), device=dev)

    def fwd_bwd():
        bucket.zero()
        loss = torch.nn.functional.cross_entropy(model(x), lab)
        loss.backward()
        bucket.pack()
        loss_buf.copy_(loss.detach())

    def eager_step():
        fwd_bwd()
        bucket.all_reduce_mean()
        opt.step()

    _log('model built; eager warm-up')
    # warm-up (also the side-stream warm-up HIP graph capture needs)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(max(2, args.warmup)):
            eager_step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()

    _log('warm-up done')
    mode = 'eager'
    step = eager_step
    if not args.no_graph:
        try:
            g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                fwd_bwd()
            with torch.cuda.graph(g2):
                opt.step()

            def graph_step():
                g1.replay()
                bucket.all_reduce_mean()
                g2.replay()
            step, mode = graph_step, 'hipgraph'
            _log('graphs captured')
            for _ in range(2):
                step()
            torch.cuda.synchronize()
        except Exception as e:                            # noqa: BLE001
            if rank == 0:
                print(f'[bench] graph capture failed ({type(e).__name__}: {e}); timing eager launches', file=sys.stderr)
            step, mode = eager_step, 'eager'
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    final_loss = float(loss_buf)
    _log(f'timed {args.steps} steps in {dt:.3f}s ({mode})')

    out = None
    # Per-kernel durations for the roofline: the same step, launched eagerly on ONE stream, HIP events around every ABI
    # launch.  (With the side streams on, kernels overlap and an event pair measures the overlap, not the kernel:
    # rocprofv3 --kernel-trace of that mode reports the same inflated durations.)
    from tam_gcn_amd import functional as _F
    side, _F.USE_SIDE_STREAMS = _F.USE_SIDE_STREAMS, False
    agg = instrumented_pass(eager_step, probe, 2)         # every rank: the step contains a collective
    _F.USE_SIDE_STREAMS = side
    _log('instrumented pass done')
    if rank == 0:
        roof, shares = roofline_of(agg)
        cpu = None if args.no_cpu_baseline or world > 1 else cpu_baseline()
        out = {
            'metric': 'skeleton clips/sec (fwd+bwd), N-UCLA 20-joint x 64-frame',
            'value': world * B * args.steps / dt, 'unit': 'clips/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'N-UCLA joint stream, {B} clips/GPU x (3,64,20,1), models.ctrgcn.Model '
                                   f'fwd+CE+bwd+grad-allreduce+SGD step, train-mode BN',
                       'global_batch': world * B, 'parallelism': f'dp{world}', 'launch': mode,
                       'final_loss': final_loss},
            'roofline': roof, 'cpu_baseline': cpu, 'abi_ms_per_2_steps': shares,
        }
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == '__main__':
    main()
