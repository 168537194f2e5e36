"""Summarise a rocprofv3 kernel-trace CSV: consecutive dispatches of the same (kernel, grid) are one
group (kbench issues each op several times back to back); prints the group's median duration."""
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
groups = []
for r in rows:
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    name = name.split('(')[0]
    key = (name, r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'], r['Workgroup_Size_X'])
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if groups and groups[-1][0] == key:
        groups[-1][1].append(d)
    else:
        groups.append((key, [d], r['LDS_Block_Size'], r['VGPR_Count'], r['Accum_VGPR_Count'], r['Scratch_Size']))
for key, ds, lds, vg, ag, sc in groups:
    if 'at::native' in key[0] or 'rocclr' in key[0]:
        continue
    wg = int(key[4])
    grid = (int(key[1]) // wg, key[2], key[3])
    print(f'{key[0][:52]:52s} x{len(ds):<3d} med {statistics.median(ds):9.1f} us  grid {grid} wg {wg} lds {lds} vgpr {vg}+{ag} scratch {sc}')
