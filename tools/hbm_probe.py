"""GPU-box tool: what this box's HBM delivers to plain streaming kernels (torch copy / add / sum) --
the practical ceiling the roofline fractions of the memory-bound kernels should be read against."""
import torch
dev = torch.device('cuda:0')
def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for mb in (84, 336, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    x = torch.randn(n, device=dev); y = torch.empty_like(x); z = torch.randn(n, device=dev)
    us = timeit(lambda: y.copy_(x)); print(f'copy  {mb:5d} MB: {us:8.1f} us  {2 * n * 4 / us / 1e6:6.2f} TB/s')
    us = timeit(lambda: torch.add(x, z, out=y)); print(f'add   {mb:5d} MB: {us:8.1f} us  {3 * n * 4 / us / 1e6:6.2f} TB/s')
    us = timeit(lambda: x.sum()); print(f'sum   {mb:5d} MB: {us:8.1f} us  {n * 4 / us / 1e6:6.2f} TB/s')
    us = timeit(lambda: y.fill_(1.0)); print(f'fill  {mb:5d} MB: {us:8.1f} us  {n * 4 / us / 1e6:6.2f} TB/s')
