"""Print PMC counters per (kernel, grid) from rocprofv3 counter_collection CSVs under the given dirs."""
import collections, csv, glob, re, sys
tab = collections.defaultdict(dict)
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r'\(.*$', '', r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', ''))
            if 'at::' in name or 'rocclr' in name or 'reduce' in name:
                continue
            key = (name[:40], r['Grid_Size'])
            tab[key][r['Counter_Name']] = float(r['Counter_Value'])
for key, c in tab.items():
    wc = c.get('SQ_WAVE_CYCLES', 0) or 1
    print(f'{key[0]:40s} grid {key[1]:>9s}')
    print('   ' + '  '.join(f'{k}={v:.3g}' for k, v in sorted(c.items())))
    if 'SQ_WAIT_ANY' in c:
        print(f'   wait_any {c["SQ_WAIT_ANY"] / wc:.0%}  wait_inst {c.get("SQ_WAIT_INST_ANY", 0) / wc:.0%}  active {c.get("SQ_ACTIVE_INST_ANY", 0) / wc:.0%}'
              f'  valu-active {c.get("SQ_ACTIVE_INST_VALU", 0) / wc:.0%}  lds-active {c.get("SQ_ACTIVE_INST_LDS", 0) / wc:.0%}')
