"""GPU-box tool: error of the GEMM kernels against an fp64 reference, for the current TAMGCN_SPLIT_BF16 mode
(0 = fp32-input MFMA, 2 = split-fp32 on the bf16 matrix cores).  Run once per mode and compare."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tam_gcn_amd import ops
from tam_gcn_amd.ops import S
dev = torch.device('cuda:0')
torch.manual_seed(0)
print('TAMGCN_SPLIT_BF16 =', os.environ.get('TAMGCN_SPLIT_BF16', '(default 1)'))
def rep(name, got, ref):
    d = (got.double() - ref).abs()
    print(f'  {name:34s} max|err|/max|ref| {float(d.max() / ref.abs().max()):.2e}   rel-L2 {float(d.norm() / ref.norm()):.2e}')
for N, M, K, T in [(64, 192, 64, 64), (64, 384, 128, 32), (64, 768, 256, 16), (64, 256, 256, 16)]:
    V = 20
    gy = torch.randn(N, M, T, V, device=dev); x = torch.randn(N, K, T, V, device=dev).relu_()
    print(f'N={N} M={M} K={K} T={T}')
    dw = ops.wgrad(S(gy), S(x), M=M, K=K).view(M, K)
    rep('wgrad', dw, torch.einsum('nmtv,nktv->mk', gy.double(), x.double()))
    w = torch.randn(M, K, 1, 1, device=dev) * (1.0 / K ** 0.5)
    y, _ = ops.conv(S(x), K=K, w=w, bias=None, M=M)
    rep('conv fwd', y, torch.einsum('mk,nktv->nmtv', w[:, :, 0, 0].double(), x.double()))
    dx, _ = ops.conv(S(gy), K=M, w=w, bias=None, M=K, wmode=1)
    rep('conv bwd-data', dx, torch.einsum('mk,nmtv->nktv', w[:, :, 0, 0].double(), gy.double()))
