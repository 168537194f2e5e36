"""Achieved HBM rate of every kernel of the bench run: PMC traffic per launch (tools/pmc_traffic.py output) over the one-stream
rocprofv3 average (tools/profile_round.sh: <tag>_bench_kernel_stats_serial.csv).   python tools/rate_table.py profiles r04e"""
import csv, re, sys
d, tag = sys.argv[1], sys.argv[2]
tr = {}
for l in open(f'{d}/{tag}_pmc_traffic.txt'):
    m = re.match(r'(.+?)\s+([\d.]+) MB/launch', l)
    if m: tr[m.group(1).strip().replace(' >', '>')] = float(m.group(2))
rows = []
for r in csv.DictReader(open(f'{d}/{tag}_bench_kernel_stats_serial.csv')):
    n = re.sub(r'\(.*$', '', r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')).strip()
    rows.append((n, int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['TotalDurationNs'])))
tot = sum(x[3] for x in rows)
print('%-56s %6s %9s %8s %7s %6s' % ('kernel', 'calls', 'avg us', 'MB', 'TB/s', '%time'))
for n, c, avg, t in sorted(rows, key=lambda x: -x[3]):
    mb = tr.get(n.replace(' >', '>'))
    print('%-56s %6d %9.1f %8s %7s %6.2f' % (n[:56], c, avg, '%.1f' % mb if mb else '-', '%.2f' % (mb / avg) if mb else '-', 100 * t / tot))
