#!/bin/bash
# GPU-box script: per-kernel time of the NTU configuration (V = 25, T = 300, 2 persons), batch 16
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/prof_ntu
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ntu -- python3 tools/config_bench.py 16 > gpurun_out/ntu.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_ntu/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f}%  x{r['Calls']:>6s}  avg {float(r['AverageNs'])/1e3:9.1f} us  {n[:70]}")
PY
rm -rf gpurun_out/prof_ntu
tail -2 gpurun_out/ntu.log
