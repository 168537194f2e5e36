"""Per-step kernel breakdown from a rocprofv3 kernel-trace CSV of bench.py:
    python tools/step_breakdown.py <kernel_trace.csv> <steps_traced|auto> [top]
groups dispatches by (kernel, grid) and prints total ms per step, launches per step and mean us."""
import csv
import collections
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
agg = collections.OrderedDict()
byname = collections.Counter()
for r in rows:
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    wg = max(int(r['Workgroup_Size_X']), 1)
    key = (name, int(r['Grid_Size_X']) // wg)
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += d
    byname[name] += d
if steps == 'auto':      # ten CTRGC forward launches per training step
    steps = sum(c for (n, g), (c, d) in agg.items() if n.startswith('ctrgc_fwd_kernel')) / 10.0
steps = float(steps)
tot = sum(v[1] for v in agg.values())
print(f'total kernel time per step: {tot / steps / 1e3:.2f} ms over {sum(v[0] for v in agg.values()) / steps:.0f} launches')
print('--- by kernel')
for n, d in byname.most_common(25):
    print(f'{n[:70]:70s} {d / steps / 1e3:8.3f} ms/step')
print('--- by (kernel, grid)')
for (n, g), (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f'{n[:56]:56s} grid {g:7d} x{c / steps:6.1f}/step  mean {d / c:8.1f} us  {d / steps / 1e3:7.3f} ms/step')
