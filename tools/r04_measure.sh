#!/bin/bash
# GPU-box script: the round-4 measurement set (copy what should be judged from gpurun_out/ into profiles/).
#   bash tools/r04_measure.sh [round|configs|all] [tag]        (tag: prefix of the output files, default r04; the end-of-round set is r04c)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
WHAT=${1:-all}
TAG=${2:-r04}
if [ "$WHAT" = round ] || [ "$WHAT" = all ]; then
  bash tools/profile_round.sh $TAG final > $O/${TAG}_profile_round.log 2>&1; tail -4 $O/${TAG}_profile_round.log
  timeout -k 10 300 python tools/roofline_table.py > $O/${TAG}_roofline_table_nucla.txt 2>&1; head -3 $O/${TAG}_roofline_table_nucla.txt
fi
if [ "$WHAT" = configs ] || [ "$WHAT" = all ]; then
  for m in 0 1; do
    TAMGCN_SPLIT_BF16=$m timeout -k 10 200 python tools/config_bench.py ntu 128 2>/dev/null | tail -1 > $O/${TAG}_config3_ntu128_mode$m.json
    TAMGCN_SPLIT_BF16=$m timeout -k 10 200 python tools/config_bench.py syn 128 2>/dev/null | tail -1 > $O/${TAG}_config4_syn128_mode$m.json
    TAMGCN_SPLIT_BF16=$m timeout -k 10 200 python tools/config_bench.py ucla52 2>/dev/null | tail -1 > $O/${TAG}_config1_ucla52_mode$m.json
  done
  timeout -k 10 300 python tools/config_bench.py syn 256 2>/dev/null | tail -1 > $O/${TAG}_config4_syn256_mode0.json
  timeout -k 10 300 python bench.py --config 4stream --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_config2_4stream.json
  for f in $O/${TAG}_config*.json; do echo "$f: $(cut -c1-200 $f)"; done
  bash tools/pmc_kernel.sh ctrgc64 python3 tools/kctrgc_only.py 64 64 64 > $O/${TAG}_pmc_ctrgc_64_64_T64.txt 2>&1
  bash tools/pmc_kernel.sh ctrgc256 python3 tools/kctrgc_only.py 256 256 16 > $O/${TAG}_pmc_ctrgc_256_256_T16.txt 2>&1
  grep -A16 "ctrgc_fwd_kernel" $O/${TAG}_pmc_ctrgc_64_64_T64.txt | head -18
  timeout -k 10 200 python tools/infer_bench.py > $O/${TAG}_infer_bench.txt 2>&1; tail -8 $O/${TAG}_infer_bench.txt
fi
