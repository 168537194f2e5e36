"""What each part of the LDS-DMA weight-gradient kernel costs at the dW3 shapes of the C = 256 layers: side builds of conv.hip with
-DTG_WKO=<mask> (results wrong by design), each in a child process.   here: python tools/wgrad_knockout.py build    box: python tools/wgrad_knockout.py
(Round 4's run, which also carried a 256 x 256 tile that was rejected: profiles/r04_wgrad_knockout.txt.)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import torch
    from tam_gcn_amd import ops
    from tam_gcn_amd.ops import S
    torch.manual_seed(0)
    dev = 'cuda'
    for nm, N, M, K, T, V in [('ucla 768x256 T16 V20 N256', 256, 768, 256, 16, 20), ('ntu 768x256 T75 V25 N128', 128, 768, 256, 75, 25),
                              ('cfg4 768x256 T128 V64 N64', 64, 768, 256, 128, 64), ('512x256 T16 V20', 256, 512, 256, 16, 20)]:
        gy = torch.randn(N, M, T, V, device=dev); x = torch.randn(N, K, T, V, device=dev)
        ref = torch.einsum('nmtv,nktv->mk', gy.double(), x.double())
        out = ops.wgrad(S(gy), S(x), M=M, K=K).view(M, K)
        err = float((out.double() - ref).abs().max() / ref.abs().max())
        for _ in range(3): ops.wgrad(S(gy), S(x), M=M, K=K)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): ops.wgrad(S(gy), S(x), M=M, K=K)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 20
        print(f'  {nm:32s} {us:9.1f} us  {2.0*N*M*K*T*V/us/1e6:7.1f} TFLOP/s  rel err {err:.2e}', flush=True)
    sys.exit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIDE = os.path.join(ROOT, 'tools', '_side')
MASKS = [1, 2, 4, 8, 6, 7]
if len(sys.argv) > 1 and sys.argv[1] == 'build':       # knock-out side builds (-DTG_WKO=mask; results wrong by design)
    from tam_gcn_amd import build as B
    B.build()
    os.makedirs(SIDE, exist_ok=True)
    csrc = os.path.join(ROOT, 'tam_gcn_amd', 'csrc')
    others = [os.path.splitext(s)[0] + '.o' for s in B.sources() if os.path.basename(s) != 'conv.hip']
    for old in os.listdir(SIDE):
        os.remove(os.path.join(SIDE, old))
    procs = []
    for m in MASKS:
        o = os.path.join(SIDE, f'conv_wko{m}.o')
        procs.append(subprocess.Popen([B._hipcc(), f'--offload-arch={B.ARCH}', '-O3', '-std=c++17', '-fPIC', f'-DTG_WKO={m}', '-c',
                                       os.path.join(csrc, 'conv.hip'), '-o', o]))
        if len(procs) == 4:
            for p in procs: assert p.wait() == 0
            procs = []
    for p in procs: assert p.wait() == 0
    for m in MASKS:
        subprocess.check_call([B._hipcc(), f'--offload-arch={B.ARCH}', '-shared', '-fPIC', '-o', os.path.join(SIDE, f'libtamgcn_wko{m}.so'),
                               os.path.join(SIDE, f'conv_wko{m}.o')] + others)
        os.remove(os.path.join(SIDE, f'conv_wko{m}.o'))
    print('built', sorted(os.listdir(SIDE)))
    sys.exit(0)
kos = [m for m in MASKS if os.path.exists(os.path.join(SIDE, f'libtamgcn_wko{m}.so'))]
for ko in [0] + kos:
    print(f'knock-out mask {ko} (1 one MFMA in eight, 2 no DMA, 4 no fragment reads, 8 no barrier)', flush=True)
    env = dict(os.environ, TAMGCN_SPLIT_BF16='0')
    if ko:
        env['TAMGCN_LIB'] = os.path.join(SIDE, f'libtamgcn_wko{ko}.so')
    subprocess.run([sys.executable, __file__, 'child'], env=env, check=True)
