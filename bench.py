#!/usr/bin/env python
"""bench.py — skeleton clips/sec (fwd + CE loss + bwd [+ gradient all-reduce] + SGD step)
of models.ctrgcn.Model on the N-UCLA joint stream (20 joints x 64 frames), BASELINE.json
configs[1]: batch 256 per GPU, synthetic U(-1,1) clips, seeded init (de-degenerated so
every branch of the block does real work).

    python bench.py [--gpus N --steps K --warmup W] [--config ucla|4stream]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no launcher environment starts N fresh rank processes itself
(torch.distributed.run, rendezvous on 127.0.0.1) BEFORE anything touches the GPU, forwards rank 0's JSON
line and exits with the children's code.  --config 4stream is BASELINE.json configs[2]: four Models
(joint / bone / joint-motion / bone-motion, derived on the GPU from the joint clips), 128 clips per GPU
and stream, ONE concatenated gradient bucket (4 x 6.77 MB) and one all-reduce per step.

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
import contextlib
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
BF16_MFMA_PEAK = 2.5e15     # FLOP/s, dense bf16 MFMA (MI355X_MICROARCH.md; the headline figures with 2:1 sparsity are not used)

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
                                                     'tamgcn_tconv_supported', 'tamgcn_tconv_nparts', 'tamgcn_tconv_wgrad_max_split',
                                                     'tamgcn_ew_nparts', 'tamgcn_ctrgc_lds_bytes', 'tamgcn_get_split_mode',
                                                     'tamgcn_set_split_mode', 'tamgcn_wgrad_max_split', 'tamgcn_ctrgc_tiled_supported',
                                                     'tamgcn_ctrgc_tiled_chunks'):
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
            self.records.append((sym or name, e0, e1, _algorithmic(name, args), _layer_tag(name, args)))
            return rc
        return wrapped


def _src_reads(s):
    return 1 + (1 if s.x2 else 0)


def _layer_tag(name, args):
    """'Cin->Cout,T' of a fused-CTRGC forward launch (the per-layer roofline table), else None."""
    if name not in ('tamgcn_ctrgc_fwd', 'tamgcn_ctrgc_tiled_agg_fwd'):
        return None
    d = args[0]._obj
    return f'{d.Cin}->{d.Cout},T{d.T},V{d.V}'


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
    if name == 'tamgcn_ctrgc_tiled_agg_fwd':       # x3 (S*Cout) in, y out, E once; the dense W3.x GEMM is a tamgcn_conv launch of its own
        b = d.N * d.T * d.V * (d.S * d.Cout + d.Cout) + d.N * d.S * d.Cout * d.V * d.V
        return (4.0 * b, 2.0 * d.N * d.S * d.Cout * d.T * d.V * d.V)
    if name == 'tamgcn_ctrgc_tiled_agg_bwd':
        b = d.N * d.T * d.V * (d.S * d.Cout + 2 * d.Cout) + d.N * d.S * d.Cout * d.V * d.V
        return (4.0 * b, 2.0 * d.N * d.S * d.Cout * d.T * d.V * d.V)
    if name == 'tamgcn_ctrgc_tiled_de_acc':
        b = d.N * d.T * d.V * (d.S * d.Cout + 2 * d.Cout) + d.N * d.S * d.Cout * d.V * d.V
        return (4.0 * b, 2.0 * d.N * d.S * d.Cout * d.T * d.V * d.V)
    if name in ('tamgcn_tconv_fwd', 'tamgcn_tconv_bwd', 'tamgcn_tconv_wgrad'):
        # the MS-TCN second stage in one launch (csrc/tconv.hip): nb temporal branches of Cb channels (+ the pooled branch in the
        # forward); DESIGN.md section 3: forward reads its source slice and writes its output slice once, the data gradient reads
        # the two-source gradient and the ReLU-mask operand and writes once, the weight gradient reads three tensors
        nbr = d.nb + (1 if (name == 'tamgcn_tconv_fwd' and d.pool) else 0)
        if name == 'tamgcn_tconv_fwd':
            b = d.N * nbr * d.Cb * (d.T_in + d.T_out) * d.V
        elif name == 'tamgcn_tconv_bwd':
            b = d.N * d.nb * d.Cb * (2 * d.T_out + 2 * d.T_in) * d.V
        else:
            b = d.N * d.nb * d.Cb * (2 * d.T_out + d.T_in) * d.V
        return (4.0 * b, 2.0 * d.N * d.nb * d.Cb * d.Cb * d.KT * d.T_out * d.V)
    if name == 'tamgcn_ctrgc_bwd_dx3':
        b = d.N * d.T * d.V * (2 * d.Cout + d.S * d.Cout)
        f = d.N * d.S * (2.0 * d.R * d.Cout * d.V * d.V + 2.0 * d.Cout * d.T * d.V * d.V)
        return (4.0 * b, f)
    return (0.0, 0.0)


def instrumented_pass(step_fn, probe, steps):
    probe.records, probe.on = [], True
    for _ in range(steps):
        step_fn()
    torch.cuda.synchronize()
    probe.on = False
    agg, layers = {}, {}
    for name, e0, e1, (b, f), tag in probe.records:
        ms = e0.elapsed_time(e1)
        for table, key in ((agg, name),) + (((layers, tag),) if tag else ()):
            a = table.setdefault(key, dict(calls=0, ms=0.0, bytes=0.0, flops=0.0))
            a['calls'] += 1
            a['ms'] += ms
            a['bytes'] += b
            a['flops'] += f
    return agg, layers


def ctrgc_layer_table(layers):
    """Per-layer roofline of the fused CTRGC forward (both fractions, binding roof named)."""
    rows = []
    for tag, a in layers.items():
        sec = a['ms'] * 1e-3 / a['calls']
        b, f = a['bytes'] / a['calls'], a['flops'] / a['calls']
        t_h, t_m = b / HBM_PEAK, f / F32_MFMA_PEAK
        rows.append(dict(layer=tag, launches=a['calls'], avg_us=round(sec * 1e6, 1), hbm_frac=round(b / sec / HBM_PEAK, 4),
                         mfma_f32_frac=round(f / sec / F32_MFMA_PEAK, 4), binding='mfma' if t_m > t_h else 'hbm',
                         frac_of_binding=round(max(t_h, t_m) / sec, 4)))
    return rows


def roofline_of(agg):
    """Roofline object for the kernel symbol with the largest share of step time (symbols as rocprofv3
    prints them, minus the anonymous-namespace prefix and the argument list)."""
    timed = {k: v for k, v in agg.items() if v['bytes'] > 0}
    if not timed:
        return None, {}
    order = sorted(timed, key=lambda k: -timed[k]['ms'])
    tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
    tj = json.load(open(tpath)) if os.path.exists(tpath) else {}

    def describe(name):
        a = timed[name]
        sec = a['ms'] * 1e-3
        # the split GEMMs spend three bf16 MFMAs per fp32 product: their matrix roof is the dense bf16 peak / 3
        split = 'split_kernel<2' in name or ', split,' in name
        mpeak = BF16_MFMA_PEAK / 3.0 if split else F32_MFMA_PEAK
        bw, fl = a['bytes'] / sec, a['flops'] / sec
        t_hbm, t_mfma = a['bytes'] / HBM_PEAK, a['flops'] / mpeak
        # HBM bytes per launch come from rocprofv3 --pmc passes (FETCH_SIZE doubled per the gfx950 note, WRITE_SIZE), which
        # cannot run inside this process: the figure is the committed one of the named profile, labelled with its source,
        # and null when that profile has no row for this kernel symbol.
        t = tj.get(name)
        traffic = t.get('hbm_bytes_per_launch') if isinstance(t, dict) else t
        traffic_src = None
        if traffic is not None:
            traffic_src = 'profiles/traffic.json (' + str(tj.get('_source', 'rocprofv3 --pmc, earlier run of this command')) + ')'
        common = dict(kernel=name, launches=a['calls'], avg_launch_us=1e3 * a['ms'] / a['calls'], traffic_source=traffic_src,
                      algorithmic_bytes_per_launch=a['bytes'] / a['calls'], algorithmic_flops_per_launch=a['flops'] / a['calls'],
                      hbm_frac=bw / HBM_PEAK, mfma_frac=fl / mpeak, mfma_peak_tflops=mpeak / 1e12,
                      mfma_dtype='bf16 x3 (two-term split of fp32 operands)' if split else 'f32', traffic=traffic)
        if not split:
            common['mfma_f32_frac'] = fl / F32_MFMA_PEAK
        # achieved / peak / frac are against the NEARER of the two a-priori roofs (the one the launch's algorithmic bytes and
        # flops say binds).  A launch that reaches less than half of BOTH is bound by neither: DESIGN.md section 3 (ring probe,
        # phase stamps) shows those kernels limited by instruction issue and per-chunk latency, and the line says so.
        if t_mfma > t_hbm:
            r = dict(bound='mfma', achieved=fl / 1e12, peak=mpeak / 1e12, unit='TFLOP/s', frac=fl / mpeak)
        else:
            r = dict(bound='hbm', achieved=bw / 1e9, peak=HBM_PEAK / 1e9, unit='GB/s', frac=bw / HBM_PEAK)
        r['nearest_roof'] = r['bound']
        # `bound` stays one of the contract's two values; a launch under half of both roofs is marked `limited_by: "issue"`
        r['limited_by'] = 'issue' if max(bw / HBM_PEAK, fl / mpeak) < 0.5 else r['bound']
        r.update(common)
        return r

    # `roofline` = the kernel symbol with the largest share of the step IN THE HEADLINE'S GEMM MODE (the instrumented steps
    # run in that mode), against its nearer roof; `north_star_kernel` carries the same object for the fused CTRGC forward --
    # the kernel BASELINE.json's north_star names -- whatever its rank, and roofline.top3 lists the three leaders.
    r = describe(order[0])
    # (round 4: the forward runs as two symbols -- E from L2 with register-staged operands at Cin < 256, ctrgc_fwd2_kernel with
    # LDS-DMA operands at Cin >= 256 -- whose launches are summed into one object; `symbols` lists them)
    fw = [k for k in order if k.startswith('ctrgc_fwd')]
    if len(fw) > 1:
        merged = ' + '.join(fw)
        timed[merged] = {f: sum(timed[k][f] for k in fw) for f in ('calls', 'ms', 'bytes', 'flops')}
        r['north_star_kernel'] = dict(describe(merged), symbols=fw)
    else:
        r['north_star_kernel'] = describe(fw[0]) if fw else None
    top = []
    for k in order[:3]:
        v = timed[k]
        sk = v['ms'] * 1e-3
        # the split GEMMs spend three bf16 MFMAs per fp32 product: their matrix roof is the dense bf16 peak / 3
        peak = BF16_MFMA_PEAK / 3.0 if ('split_kernel<2' in k or ', split,' in k) else F32_MFMA_PEAK
        fh, fm = v['bytes'] / sk / HBM_PEAK, v['flops'] / sk / peak
        near = 'mfma' if v['flops'] / peak > v['bytes'] / HBM_PEAK else 'hbm'
        top.append(dict(kernel=k, ms=round(v['ms'], 3), launches=v['calls'], bound=near, limited_by=near if max(fh, fm) >= 0.5 else 'issue', nearest_roof=near,
                        hbm_frac=round(fh, 4), mfma_frac=round(fm, 4), mfma_peak_tflops=round(peak / 1e12, 1)))
    r['top3'] = top
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


def cpu_baseline():
    """The oracle (kind 'port': our stock-PyTorch restatement, pinned to the reference by tests/golden) on the host
    cores, forward + CE + backward (optimizer step excluded).  Legs: the reference's batch size (16,
    config/nucla/gcn.yaml:37) at T = 64 and at its real clip length (T = 52, feeder/feeder_nucla_gcn.py:26), and the GPU
    line's own batch (256, T = 64); `value` is the fastest T = 64 leg.  Bounded: fixed step counts, ~1.5 min on a 16-thread share."""
    from oracle import ctrgcn_oracle as O
    from tam_gcn_amd.models.ctrgcn import Model
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m = Model(**MODEL_ARGS)
    dedegenerate_(m)
    sd = O.clone_state(m.state_dict(), requires_grad=True)
    params = [v for v in sd.values() if v.requires_grad]

    def leg(B, T, warm, timed):
        g = torch.Generator().manual_seed(1234)
        x = torch.rand(B, 3, T, V_JOINTS, 1, generator=g) * 2 - 1
        lab = torch.randint(0, 10, (B,), generator=g)

        def step():
            for p in params:
                p.grad = None
            torch.nn.functional.cross_entropy(O.model_forward(x, sd, 20, training=True), lab).backward()

        for _ in range(warm):
            step()
        t0 = time.perf_counter()
        for _ in range(timed):
            step()
        dt = time.perf_counter() - t0
        _log(f'cpu baseline: B={B} T={T}: {timed} steps in {dt:.1f}s after {warm} warm-up on {cores} threads')
        return dict(batch=B, T=T, clips_per_s=B * timed / dt, steps=timed, warmup=warm, seconds=round(dt, 2))

    # 3 warm-up + >= 5 timed steps at the reference's batch size; the bench workload's own batch (256)
    # costs ~17 s per step on a 16-thread share, so it gets 2 warm-up + 3 timed (what bounds this function's run time)
    legs = [leg(16, T_FRAMES, 3, 8), leg(16, 52, 3, 8), leg(PER_GPU_BATCH, T_FRAMES, 2, 3)]
    # `value` = the CPU's BEST leg at the GPU line's clip length (T = 64): the fair figure to set a GPU rate against (the
    # batch-256 leg is ~5x slower per clip on the host: cache footprint); every leg stays in `legs`
    main_leg = max((l for l in legs if l['T'] == T_FRAMES), key=lambda l: l['clips_per_s'])
    return dict(value=main_leg['clips_per_s'], unit='clips/s', cores=torch.get_num_threads(), kind='port',
                sample=f"best T=64 leg: {main_leg['steps']} steps of batch {main_leg['batch']} (T=64,V=20) fwd+CE+bwd after "
                       f"{main_leg['warmup']} warm-up, {main_leg['seconds']}s; optimizer step excluded; all legs in `legs`",
                legs=legs)


# ---------------------------------------------------------------------------
# the N-UCLA bone table as a parent array (reference feeder/feeder_nucla_gcn.py:27-28: pair (v1, v2) at list
# position v1 - 1; 0-based parent of joint v = v2 - 1)
UCLA_BONE_PARENT = [1, 2, 2, 2, 2, 4, 5, 6, 2, 8, 9, 10, 0, 12, 13, 14, 0, 16, 17, 18]
STREAMS = ('joint', 'bone', 'motion', 'bone_motion')


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _self_launch(n, argv):
    """`python bench.py --gpus N` outside a launcher: start N fresh rank processes.  Runs before this process has made
    any HIP / torch.cuda call (a process that has initialised the GPU must not be re-executed or forked from)."""
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    r = subprocess.run(cmd, env=env)                    # children inherit stdout: rank 0's JSON line is ours
    raise SystemExit(r.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=None, help='clips per GPU (per stream); default 256, 128 for --config 4stream')
    ap.add_argument('--config', choices=('ucla', '4stream'), default='ucla')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--fork-streams', type=int, default=None,
                    help='4stream: number of HIP streams the four models are spread over (forward and, through autograd, backward), '
                         'inside the HIP graph or eagerly: 4 (default) or 2 = that many at a time, 0 or 1 = one after the other on the '
                         'current stream.  Bit-identical to the one-stream step either way (tests/test_gpu_configs.py)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--overlap-allreduce', type=int, default=0, metavar='NSEG',
                    help='N > 1 ranks: exchange the gradient bucket in NSEG contiguous segments, each all-reduced on a communication '
                         'stream as soon as backward has produced its last gradient (distributed.SegmentedReducer).  Needs Python '
                         'hooks to run, so the step is launched eagerly (implies --no-graph); the default keeps the HIP-graph step '
                         'with ONE all-reduce between its two graphs (~0.15 ms of ~31 ms)')
    ap.add_argument('--gemm-mode', choices=('exact', 'split'), default='exact',
                    help="arithmetic of the headline step: 'exact' (default) = the reference's own fp32 in every GEMM "
                         "(tamgcn_set_split_mode(0)); 'split' = backward weight-gradient and C>=128 data-gradient GEMMs as a 2-term "
                         "bf16 split (mode 1; the line's dtype says so) -- for profiling that mode on its own")
    ap.add_argument('--single-mode', action='store_true',
                    help='do not measure the other GEMM mode beside the headline (one rocprofv3 stats file per mode: tools/profile_round.sh)')
    args = ap.parse_args()

    # TAMGCN_BENCH_REHEARSAL=1: the multi-rank CONTROL FLOW (spawn, rendezvous, broadcast, step, one flat all-reduce, flat SGD,
    # barriers, max-over-ranks clock, rank 0's JSON line) on CPU tensors over gloo with a stand-in gradient fill instead
    # of the HIP model: what tests/test_distributed_cpu.py drives with world size 2.  Never a measurement: `value` is null.
    rehearsal = os.environ.get('TAMGCN_BENCH_REHEARSAL') == '1'
    nfork = 0
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        _self_launch(args.gpus, sys.argv[1:])
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')
    # Rehearsal hooks for a one-GPU box (RCCL needs one device per rank): TAMGCN_BENCH_ONE_DEVICE=1 puts every rank on
    # cuda:0 and TAMGCN_DIST_BACKEND=gloo moves the collectives to gloo -- same control flow, not a measurement.
    if os.environ.get('TAMGCN_BENCH_ONE_DEVICE') == '1':
        local = 0
    if rehearsal:
        dev = torch.device('cpu')
    else:
        torch.cuda.set_device(local)
        dev = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        backend = 'gloo' if rehearsal else os.environ.get('TAMGCN_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from tam_gcn_amd.distributed import ParamArena, SGDNesterov, SegmentedReducer, broadcast_state
    if args.overlap_allreduce > 1:
        args.no_graph = True
    from tam_gcn_amd.models.ctrgcn import Model
    probe = None
    if not rehearsal:
        from tam_gcn_amd import _lib
        probe = _Probe(_lib.load())
        _lib._lib = probe                                 # every ABI launch goes through the probe
        probe._lib.tamgcn_set_split_mode(0 if args.gemm_mode == 'exact' else 1)

    four = args.config == '4stream'
    B = args.batch or (128 if four else PER_GPU_BATCH)
    streams = STREAMS if four else ('joint',)
    torch.manual_seed(0)
    models = torch.nn.ModuleList()
    for i, _ in enumerate(streams):                       # independent models, as the 4-stream recipe trains them
        m = Model(**MODEL_ARGS)
        dedegenerate_(m, seed=i)
        models.append(m)
    models = models.to(dev).train()
    arena = ParamArena(models)                            # every model's parameters in ONE flat buffer ...
    broadcast_state(models, arena=arena)                  # one broadcast of the arena + one per dtype of packed buffers
    bucket = arena.grad_bucket()                          # ... and ONE congruent gradient bucket: one all-reduce per step
    opt = SGDNesterov(arena.params, lr=0.01, momentum=0.9, weight_decay=1e-4, arena=arena, bucket=bucket)
    g = torch.Generator().manual_seed(1234 + rank)
    x = (torch.rand(B, 3, T_FRAMES, V_JOINTS, 1, generator=g) * 2 - 1).to(dev)
    lab = torch.randint(0, 10, (B,), generator=g).to(dev)
    loss_buf = torch.zeros((), device=dev)
    parent = torch.tensor(UCLA_BONE_PARENT, dtype=torch.int32, device=dev)
    reducer = None
    if args.overlap_allreduce > 1:
        reducer = SegmentedReducer(bucket, nseg=args.overlap_allreduce,
                                   comm_stream=None if rehearsal else torch.cuda.Stream(dev))

    if rehearsal:
        def fwd_bwd():
            bucket.zero()
            if reducer is not None:                       # through autograd, so that the segment hooks fire as in a real backward
                reducer.begin()
                sum((p * (1e-3 * (rank + 1) * ((i % 7) - 3))).sum() for i, p in enumerate(arena.params)).backward()
            else:
                for i, p in enumerate(arena.params):      # stand-in gradients: deterministic, rank-dependent
                    p.grad = torch.full_like(p, 1e-3 * (rank + 1) * ((i % 7) - 3))
                bucket.pack()
            loss_buf.fill_(float(rank))
    else:
        from tam_gcn_amd import ops as _ops, functional as _Fm
        from tam_gcn_amd.functional import CrossEntropyLoss
        ce = CrossEntropyLoss()                             # the harness's nn.CrossEntropyLoss() as two HIP launches (row f1)

        if args.fork_streams is None:
            args.fork_streams = 4 if four and not rehearsal else 0
        nfork = args.fork_streams if four and args.fork_streams > 1 else 0
        fork = nfork > 1
        pool = [torch.cuda.Stream(dev) for _ in range(nfork)]
        model_streams = [pool[i % nfork] for i in range(len(streams))] if fork else []     # models i, i + nfork, ... share a stream

        def fwd_bwd():
            bucket.zero()
            total = None
            cur = torch.cuda.current_stream(dev)
            losses = []
            for i, (name, m) in enumerate(zip(streams, models)):   # the streams are derived on the GPU from the resident joint clips
                ctx = contextlib.nullcontext()
                if fork:                                  # the four models are independent until the bucket.
                    if i < nfork:
                        model_streams[i].wait_stream(cur)     # Autograd replays every node on the stream of its forward and joins
                    ctx = _Fm.model_stream(model_streams[i])     # the leaf streams into `cur` when backward() returns.
                with ctx:
                    xs = x if name == 'joint' else _ops.stream_derive(x, parent, name)
                    losses.append(ce(m(xs), lab))
            for st in pool:
                cur.wait_stream(st)
            for loss in losses:
                total = loss if total is None else total + loss
            if reducer is not None:
                reducer.begin()                           # segments leave from the gradient hooks during backward
                total.backward()
            else:
                total.backward()
                bucket.pack()
            loss_buf.copy_(total.detach() / len(streams))

    def eager_step():
        fwd_bwd()
        if reducer is not None:
            reducer.finish()
        else:
            bucket.all_reduce_mean()
        opt.step()

    def sync():
        if not rehearsal:
            torch.cuda.synchronize()

    _log('model built; eager warm-up')
    mode = 'eager'
    step = eager_step
    if rehearsal:
        for _ in range(max(1, args.warmup)):
            eager_step()
    else:
        # warm-up (also the side-stream warm-up HIP graph capture needs)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(max(2, args.warmup)):
                eager_step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        _log('warm-up done')
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
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
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
    roof, shares, layer_rows, ms_other = None, {}, [], None
    split = 0
    if not rehearsal:
        # the same step in the OTHER GEMM mode, re-captured, a few replays: reported beside the headline (exact fp32 by
        # default; the other mode is then the 2-term bf16 split in the backward GEMMs) under its own keys
        lib = probe._lib
        split = lib.tamgcn_get_split_mode()
        if world == 1 and not args.single_mode:
            lib.tamgcn_set_split_mode(1 - split)
            try:
                step0 = eager_step
                if mode == 'hipgraph':
                    g1x = torch.cuda.CUDAGraph()
                    eager_step(); torch.cuda.synchronize()
                    with torch.cuda.graph(g1x):
                        fwd_bwd()

                    def step0():
                        g1x.replay()
                        g2.replay()
                for _ in range(2):
                    step0()
                torch.cuda.synchronize()
                k0 = max(3, min(10, args.steps))
                t0 = time.perf_counter()
                for _ in range(k0):
                    step0()
                torch.cuda.synchronize()
                ms_other = 1e3 * (time.perf_counter() - t0) / k0
            finally:
                lib.tamgcn_set_split_mode(split)
            _log(f'GEMM mode {1 - split} step: {ms_other:.2f} ms')
    if not rehearsal:
        # Per-kernel durations for the roofline: the same step, launched eagerly on ONE stream, HIP events around every ABI
        # launch.  (With the side streams on, kernels overlap and an event pair measures the overlap, not the kernel:
        # rocprofv3 --kernel-trace of that mode reports the same inflated durations.)
        from tam_gcn_amd import functional as _F
        side, _F.USE_SIDE_STREAMS = _F.USE_SIDE_STREAMS, False
        agg, layers = instrumented_pass(eager_step, probe, 2)     # every rank: the step contains a collective
        _F.USE_SIDE_STREAMS = side
        _log('instrumented pass done')
        if rank == 0:
            roof, shares = roofline_of(agg)
            layer_rows = ctrgc_layer_table(layers)
    if rank == 0:
        if args.no_cpu_baseline or world > 1 or rehearsal:
            cpu = None
            cpu_why = ('--no-cpu-baseline' if args.no_cpu_baseline else 'rehearsal' if rehearsal else
                       'timed on rank 0 of the N = 1 run only (the host cores are shared by the N ranks here)')
        else:
            cpu, cpu_why = cpu_baseline(), None
        clips = world * B * args.steps * len(streams)
        other = {0: 'exact_f32', 1: 'split_bf16'}[1 - split]
        DTYPES = {0: 'f32 (exact fp32-input MFMA v_mfma_f32_16x16x4_f32 in every GEMM, forward and backward: the reference\'s arithmetic)',
                  1: 'f32 storage; fwd GEMMs exact fp32-input MFMA; bwd weight-gradient and C>=128 data-gradient GEMMs 2-term '
                     'bf16 split = 3 bf16 MFMAs, ~4.5e-6 rel. error (TAMGCN_SPLIT_BF16=1, opt-in, NOT the reference\'s precision)'}
        out = {
            'metric': 'skeleton clips/sec (fwd+CE+bwd+grad all-reduce+SGD step), N-UCLA 20-joint x 64-frame',
            'value': None if rehearsal else clips / dt, 'unit': 'clips/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPES[split], 'gemm_mode': split,
            # the other GEMM mode, measured in the same run (null with --single-mode or N > 1): never the headline
            f'ms_per_step_{other}': ms_other,
            f'value_{other}': None if (ms_other is None or rehearsal) else world * B * len(streams) / (ms_other * 1e-3),
            f'dtype_{other}': DTYPES[1 - split],
            'data': 'synthetic',
            'config': {'workload': (f'N-UCLA 4-stream (joint/bone/motion/bone-motion derived on GPU), 4 x models.ctrgcn.Model, '
                                    f'{B} clips/GPU/stream x (3,64,20,1), ONE {arena.total * 4 / 1e6:.1f} MB gradient bucket' if four else
                                    f'N-UCLA joint stream, {B} clips/GPU x (3,64,20,1), models.ctrgcn.Model') +
                                   ' fwd+CE+bwd+grad-allreduce+SGD step, train-mode BN',
                       'global_batch': world * B, 'streams': len(streams), 'parallelism': f'dp{world}', 'launch': mode,
                       'model_streams': nfork, 'final_loss': final_loss, 'rehearsal': rehearsal,
                       'allreduce': (f'{len(reducer.ranges)} segments overlapped with backward' if reducer is not None else
                                     'one flat bucket after backward')},
            'roofline': roof, 'cpu_baseline': cpu, 'cpu_baseline_absent_because': cpu_why, 'ctrgc_fwd_layers': layer_rows,
            'abi_ms_per_2_steps': shares,
        }
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
