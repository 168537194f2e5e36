"""Turn rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs) into HBM bytes per launch per kernel.
usage: pmc_traffic.py <dir_fetch> <dir_write> <out.json> [source label, e.g. "r03 @ commit abc1234"]
Units/corrections (MI355X_MICROARCH.md §HBM): both counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of wide (16 B/lane) coalesced streaming reads, so the read side is doubled
for kernels whose global loads are float4 (all product kernels here); WRITE_SIZE is exact for 16-B stores."""
import collections
import csv
import glob
import json
import re
import sys


def per_kernel(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
                name = re.sub(r'\(.*$', '', name)
                agg[name].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


fetch, nf = per_kernel(sys.argv[1], 'FETCH_SIZE')
write, _ = per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
    out[k] = dict(hbm_bytes_per_launch=2 * f + w, fetch_raw_bytes=f, fetch_corrected_bytes=2 * f, write_bytes=w,
                  launches=nf.get(k, 0))
if len(sys.argv) > 4:
    out['_source'] = 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `python3 bench.py --steps 2 --warmup 1`, ' + sys.argv[4]
json.dump(out, open(sys.argv[3], 'w'), indent=1)
for k, v in out.items():
    if isinstance(v, dict) and 'at::' not in k and 'rocclr' not in k:
        print(f'{k[:60]:60s} {v["hbm_bytes_per_launch"] / 1e6:10.1f} MB/launch (fetch x2 {v["fetch_corrected_bytes"] / 1e6:.1f}, write {v["write_bytes"] / 1e6:.1f}) n={v["launches"]}')
