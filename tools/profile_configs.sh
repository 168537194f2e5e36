#!/bin/bash
# GPU-box script: rocprofv3 evidence for the BASELINE configurations beside the bench line.  usage: bash tools/profile_configs.sh <tag>
#   gpurun_out/<tag>_config4_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python3 tools/config_bench.py syn 64`
#   gpurun_out/<tag>_config4_pmc_traffic.txt    FETCH_SIZE / WRITE_SIZE (separate passes) -> HBM bytes per launch of the V = 64 kernels
#   gpurun_out/<tag>_config3_kernel_stats.csv   the same kernel statistics for `python3 tools/config_bench.py ntu 32`
set -o pipefail
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
rm -rf $O/pc_stats $O/pc_fetch $O/pc_write $O/pn_stats
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pc_stats -- python3 tools/config_bench.py syn 64 > $O/${TAG}_config4_under_rocprof.log 2>&1 || exit 2
cp $(ls $O/pc_stats/*/*kernel_stats.csv | head -1) $O/${TAG}_config4_kernel_stats.csv
echo "config4 stats done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pc_fetch -- python3 tools/config_bench.py syn 32 > $O/pcf.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pc_write -- python3 tools/config_bench.py syn 32 > $O/pcw.log 2>&1 || exit 4
python tools/pmc_traffic.py $O/pc_fetch $O/pc_write $O/${TAG}_config4_traffic.json > $O/${TAG}_config4_pmc_traffic.txt || exit 5
echo "config4 pmc done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pn_stats -- python3 tools/config_bench.py ntu 32 > $O/${TAG}_config3_under_rocprof.log 2>&1 || exit 6
cp $(ls $O/pn_stats/*/*kernel_stats.csv | head -1) $O/${TAG}_config3_kernel_stats.csv
rm -rf $O/pc_stats $O/pc_fetch $O/pc_write $O/pn_stats
grep -i "agg\|de_acc\|E_tiled\|tail_tiled" $O/${TAG}_config4_pmc_traffic.txt
