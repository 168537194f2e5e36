"""GPU-box tool: the BASELINE.json configurations that are not the bench line (bench.py measures configs[1]).

    python tools/config_bench.py ucla52            N-UCLA at the reference's real clip length T = 52, batch 256
    python tools/config_bench.py ntu [batch]       configs[3]: NTU-RGB+D 25 joints x 300 frames x 2 persons (default batch 128)
    python tools/config_bench.py syn [clips]       configs[4]: ONE TCN_GCN_unit(256, 256) on (clips, 256, 512, 64), synthetic
                                                   64-joint graph (default 256 clips = one GPU's share of batch 2048 over 8)

Every mode prints one JSON line: ms per step (fwd + loss + bwd, eager launches), clips/s, peak memory, and the
per-kernel table of one instrumented step (HIP events around every ABI launch, side streams off) with the
algorithmic bytes / flops of the CTRGC kernels against the HBM and fp32-MFMA roofs."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B                                                   # noqa: E402  (the ABI probe and roofline helpers)
from tam_gcn_amd import _lib, functional as Fn                      # noqa: E402
from tam_gcn_amd.models.ctrgcn import Model, TCN_GCN_unit          # noqa: E402
from tam_gcn_amd.graph import synthetic                             # noqa: E402

dev = torch.device('cuda:0')
probe = B._Probe(_lib.load())
_lib._lib = probe


def timed(step, n=5, warm=2):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def kernel_table(step):
    side, Fn.USE_SIDE_STREAMS = Fn.USE_SIDE_STREAMS, False
    agg, layers = B.instrumented_pass(step, probe, 1)
    Fn.USE_SIDE_STREAMS = side
    rows = []
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]['ms']):
        sec = a['ms'] * 1e-3
        # the split GEMMs spend three bf16 MFMAs per fp32 product: their matrix roof is the dense bf16 peak / 3, not the
        # fp32-input MFMA peak (against which they would read > 1)
        split = 'split_kernel<2' in k or ', split,' in k
        peak = B.BF16_MFMA_PEAK / 3.0 if split else B.F32_MFMA_PEAK
        rows.append(dict(kernel=k, launches=a['calls'], ms=round(a['ms'], 3),
                         hbm_frac=round(a['bytes'] / sec / B.HBM_PEAK, 4) if a['bytes'] else None,
                         mfma_frac=round(a['flops'] / sec / peak, 4) if a['flops'] else None,
                         mfma_peak_tflops=round(peak / 1e12, 1)))
    return rows, B.ctrgc_layer_table(layers)


def model_case(name, margs, shape):
    torch.manual_seed(0)
    m = Model(**margs)
    B.dedegenerate_(m)
    m = m.to(dev).train()
    x = (torch.rand(*shape, device=dev) * 2 - 1)
    lab = torch.randint(0, margs['num_class'], (shape[0],), device=dev)

    def step():
        for p in m.parameters():
            p.grad = None
        torch.nn.functional.cross_entropy(m(x), lab).backward()

    dt = timed(step)
    rows, layers = kernel_table(step)
    print(json.dumps(dict(config=name, shape=list(shape), ms_per_step=round(dt * 1e3, 2), clips_per_s=round(shape[0] / dt, 1),
                          peak_mem_gib=round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), launch='eager',
                          ctrgc_fwd_layers=layers, kernels=rows[:24])), flush=True)


def syn_case(clips, T=512, V=64, C=256):
    torch.manual_seed(0)
    A = synthetic.Graph(num_node=V).A
    blk = TCN_GCN_unit(C, C, A)
    with torch.no_grad():
        blk.gcn1.alpha.fill_(0.5)
        blk.gcn1.bn.weight.fill_(1.0)
        w = blk.gcn1.offset_conv[0].weight
        w.copy_(torch.randn(w.shape) * (2.0 / w.shape[0]) ** 0.5)
    blk = blk.to(dev).train()
    x = (torch.rand(clips, C, T, V, device=dev) * 2 - 1).requires_grad_(True)
    cot = torch.rand(clips, C, T, V, device=dev) * 2 - 1

    def step():
        for p in blk.parameters():
            p.grad = None
        x.grad = None
        blk(x).backward(cot)

    dt = timed(step, n=3, warm=1)
    rows, layers = kernel_table(step)
    flops_fwd = clips * (16.31e9 + 357.0e6 * (C / 64) ** 2 * (T * V) / (64 * 20) * 0)      # fused-CTRGC part; rest reported per kernel
    print(json.dumps(dict(config='syn V=64 T=512 C=256 TCN_GCN_unit(256,256)', shape=[clips, C, T, V], ms_per_step=round(dt * 1e3, 1),
                          clips_per_s=round(clips / dt, 2), peak_mem_gib=round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                          ctrgc_fwd_gflop_per_clip=16.31, launch='eager', ctrgc_fwd_layers=layers, kernels=rows[:30])), flush=True)
    del flops_fwd


if __name__ == '__main__':
    mode = sys.argv[1] if len(sys.argv) > 1 else 'ucla52'
    if mode == 'ucla52':
        model_case('N-UCLA T=52', dict(num_class=10, num_point=20, num_person=1, graph='graph.ucla.Graph', graph_args=dict(labeling_mode='spatial')),
                   (256, 3, 52, 20, 1))
    elif mode == 'ntu':
        nb = int(sys.argv[2]) if len(sys.argv) > 2 else 128
        model_case('NTU-RGB+D T=300 M=2', dict(num_class=60, num_point=25, num_person=2, graph='graph.ntu_rgb_d.Graph',
                                              graph_args=dict(labeling_mode='spatial')), (nb, 3, 300, 25, 2))
    elif mode == 'syn':
        syn_case(int(sys.argv[2]) if len(sys.argv) > 2 else 256)
    else:
        raise SystemExit(__doc__)
