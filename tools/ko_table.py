"""Tabulate the output of tools/conv_knockout.py run (one column per arm)."""
import re, sys
rows, order, cols, seen, m = {}, [], [], {}, None
for l in open(sys.argv[1]):
    if l.startswith('====='):
        m = l.split()[-1].strip(); seen[m] = seen.get(m, 0) + 1; m = m + ('#%d' % seen[m] if seen[m] > 1 else ''); cols.append(m); continue
    g = re.match(r'(.{28})\s+([\d.]+) us', l)
    if g:
        nm = g.group(1).strip()
        if nm not in rows: rows[nm] = {}; order.append(nm)
        rows[nm][m] = float(g.group(2))
print('%-26s' % 'shape (us)' + ''.join('%12s' % x for x in cols))
for nm in order: print('%-26s' % nm + ''.join('%12.1f' % rows[nm].get(x, -1) for x in cols))
