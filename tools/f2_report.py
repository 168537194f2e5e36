"""GPU-box diagnostic: every block of the N-UCLA model in eval mode, fed the fp64 oracle's own input for that block, through
(a) the small-batch kernel family (tam_gcn_amd/f2.py) and (b) the general eval path; error of each against the fp64 oracle
in units of max|ref|.      python tools/f2_report.py [N T M]"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from cases import MODEL_CASES, MODEL_PARAM_SEED, MODEL_X_SEED
from params import fill_state_, make_input
from tam_gcn_amd import f2
from tam_gcn_amd.models import ctrgcn as M
from oracle import ctrgcn_oracle as O
N, T, Mp = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 52, 1)
margs = dict(MODEL_CASES[1][1], num_person=Mp)
m = M.Model(**margs)
fill_state_(m.state_dict(), seed=MODEL_PARAM_SEED)
if os.environ.get('F2_REPORT_SEEDED_STATS', '0') == '0':      # the reference's running statistics for this state (fixture), not seeded ones
    import numpy as np
    gold = np.load(os.path.join(ROOT, 'tests', 'golden', 'models.npz'))
    sd = m.state_dict()
    with torch.no_grad():
        for k in sd:
            key = f'ucla_t52/evalbuf/{k}'
            if 'running_' in k and key in gold.files and tuple(gold[key].shape) == tuple(sd[k].shape):
                sd[k].copy_(torch.from_numpy(gold[key]))
sd64 = {k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in m.state_dict().items()}
x = make_input((N, 3, T, 20, Mp), seed=MODEL_X_SEED)
h, _, _ = O._stem(x.double(), sd64, 20, False)
ins, outs = [], []
for i in range(1, 11):
    ins.append(h)
    h = O.tcn_gcn_unit(h, sd64, f'l{i}', O._STRIDES.get(i, 1), residual=(i != 1), training=False)
    outs.append(h)
m = m.cuda().eval()
eng = f2.FusedEval(m)
blocks = eng._packed(torch.device('cuda:0'))
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
os.environ['TAMGCN_F2'] = '0'
for i, (b, xin, ref) in enumerate(zip(blocks, ins, outs), 1):
    xg = xin.float().cuda().contiguous()
    got = eng._block(b, xg, st).double().cpu()
    with torch.no_grad():
        gen = getattr(m, f'l{i}')(xg).double().cpu()
    sc = float(ref.abs().max())
    e1, e2 = (got - ref).abs(), (gen - ref).abs()
    w = int(e1.argmax())
    print(f'l{i}: f2 {float(e1.max()) / sc:.3e}  general {float(e2.max()) / sc:.3e}  f2-vs-general {float((got - gen).abs().max()) / sc:.3e}   '
          f'(max|ref| {sc:.3f}; worst f2 entry: ref {float(ref.flatten()[w]):.6f} got {float(got.flatten()[w]):.6f} general {float(gen.flatten()[w]):.6f})')
