"""Which kernels sit around the __amd_rocclr_copyBuffer dispatches of a replayed step?  usage: python tools/copy_neighbours.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
short = lambda n: n.replace('(anonymous namespace)::', '').split('(')[0][-70:]
# the last complete step: between the last two SGD launches is not robust; take the last 1500 dispatches
tail = rows[-1500:]
pairs = collections.Counter(); sizes = collections.Counter()
for i, r in enumerate(tail):
    if 'copyBuffer' in r['Kernel_Name']:
        prev = next((short(tail[j]['Kernel_Name']) for j in range(i - 1, -1, -1) if 'copyBuffer' not in tail[j]['Kernel_Name']), '?')
        nxt = next((short(tail[j]['Kernel_Name']) for j in range(i + 1, len(tail)) if 'copyBuffer' not in tail[j]['Kernel_Name']), '?')
        pairs[(prev, nxt)] += 1
        sizes[(r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')))] += 1
print('copyBuffer dispatches in the last 1500:', sum(pairs.values()))
for (p, n), c in pairs.most_common(40):
    print(f'{c:4d}  after {p:55s} before {n}')
print('grid / workgroup sizes:', sizes.most_common(8))
