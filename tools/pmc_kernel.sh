#!/bin/bash
# GPU-box script: matrix-pipe / LDS / wait counters of one command's kernels (separate rocprofv3 --pmc passes).
#   bash tools/pmc_kernel.sh <tag> python3 tools/kconv_only.py --K 256 --M 768 --T 75 --V 25 --nostats --plain
set -o pipefail
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/pmc_$TAG
rm -rf $O; mkdir -p $O
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM_RD"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $SET --output-format csv -d $O/p$i -- "$@" > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/p$i.log; exit 2; }
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
tab = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(O + '/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        t = tab[k][r['Counter_Name']]
        t[0] += float(r['Counter_Value']); t[1] += 1
for k, cs in tab.items():
    if not any(x in k for x in ('conv', 'ctrgc', 'wgrad')):
        continue
    print(k)
    for c, (v, n) in sorted(cs.items()):
        print(f'    {c:34s} {v / n:16.0f} per launch  (n={n})')
PY
