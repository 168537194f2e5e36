"""GPU-box tool: which lines of the host side launch the small kernels of a training step?  One eager fwd + bwd + SGD step of the
N-UCLA model under torch.profiler (with stacks); every device kernel / memcpy shorter than `--below` us is attributed to the
innermost tam_gcn_amd (or bench-level) Python frame of the CPU op that launched it.
    python tools/small_launches.py [--batch 256] [--below 15]"""
import argparse, collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=256)
ap.add_argument('--below', type=float, default=15.0)
args = ap.parse_args()
import bench as B
from tam_gcn_amd.models.ctrgcn import Model
from tam_gcn_amd.distributed import ParamArena, SGDNesterov
from tam_gcn_amd.functional import CrossEntropyLoss
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = Model(**B.MODEL_ARGS); B.dedegenerate_(m)
models = torch.nn.ModuleList([m]).to(dev).train()
arena = ParamArena(models); bucket = arena.grad_bucket()
opt = SGDNesterov(arena.params, lr=0.01, momentum=0.9, weight_decay=1e-4, arena=arena, bucket=bucket)
x = (torch.rand(args.batch, 3, B.T_FRAMES, B.V_JOINTS, 1) * 2 - 1).to(dev)
lab = torch.randint(0, 10, (args.batch,)).to(dev)
ce = CrossEntropyLoss()

def step():
    bucket.zero()
    ce(m(x), lab).backward()
    bucket.pack()
    opt.step()

for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
# device events carry the correlation of the CPU launch; walk up the CPU op tree for a stack with one of our frames
by_line = collections.Counter(); t_line = collections.Counter(); names = collections.defaultdict(collections.Counter)
nk = 0; tk = 0.0
cpu = [e for e in ev if e.device_type == torch.autograd.DeviceType.CPU]
for e in cpu:
    ks = [k for k in e.kernels]
    if not ks: continue
    stack = None
    p = e
    while p is not None and not stack:
        stack = [f for f in (p.stack or []) if 'tam_gcn_amd' in f or 'small_launches' in f or 'bench.py' in f]
        p = p.cpu_parent
    where = stack[0].strip() if stack else '(no python frame: autograd engine / optimizer internals) ' + e.name
    for k in ks:
        us = k.duration
        if us < args.below:
            by_line[where] += 1; t_line[where] += us; names[where][k.name[:60]] += 1
            nk += 1; tk += us
print(f'{nk} device launches shorter than {args.below} us in one step, {tk / 1e3:.3f} ms of device time in total')
for w, c in by_line.most_common(60):
    print(f'{c:4d}  {t_line[w]:8.1f} us  {w[-150:]}')
    for n, cc in names[w].most_common(3):
        print(f'            {cc:3d} x {n}')

# device-to-device copies (hipMemcpyAsync nodes: torch's copy_ of a contiguous tensor): by the chain of CPU ops that asked for them
chains = collections.Counter(); sizes = collections.defaultdict(list)
for e in cpu:
    if e.name in ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::_to_copy'):
        if e.cpu_parent is not None and e.cpu_parent.name in ('aten::clone', 'aten::contiguous', 'aten::copy_', 'aten::to', 'aten::_to_copy'):
            continue                                           # count the outermost op only
        ch, p = [], e
        while p is not None and len(ch) < 4:
            ch.append(p.name); p = p.cpu_parent
        key = ' <- '.join(ch)
        chains[key] += 1
        try:
            sizes[key].append(tuple(e.input_shapes[0]) if e.input_shapes else ())
        except Exception:
            pass
print(f'\ncopy-like CPU ops in the step: {sum(chains.values())}')
for k, c in chains.most_common(25):
    print(f'{c:4d}  {k}    e.g. shapes {sizes[k][:3]}')
