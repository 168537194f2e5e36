#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
bash tools/pmc_kernel.sh ctrgc_64_64_T64 python3 tools/kctrgc_only.py 64 64 64 > $O/r03_pmc_ctrgc_64_64_T64.txt 2>&1 || { tail -5 $O/r03_pmc_ctrgc_64_64_T64.txt; exit 2; }
bash tools/pmc_kernel.sh ctrgc_256_256_T16 python3 tools/kctrgc_only.py 256 256 16 > $O/r03_pmc_ctrgc_256_256_T16.txt 2>&1 || { tail -5 $O/r03_pmc_ctrgc_256_256_T16.txt; exit 3; }
rm -rf $O/pmc_ctrgc_64_64_T64 $O/pmc_ctrgc_256_256_T16
grep -A16 "ctrgc_fwd" $O/r03_pmc_ctrgc_64_64_T64.txt | head -18
grep -A16 "ctrgc_fwd" $O/r03_pmc_ctrgc_256_256_T16.txt | head -18
rm -rf $O/prof_serial
TAMGCN_SIDE_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_serial -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/ps.log 2>&1 || exit 6
cp $(ls $O/prof_serial/*/*kernel_trace.csv | head -1) $O/r03r_bench_kernel_trace.csv
rm -rf $O/prof_serial
ls -la $O/r03r_bench_kernel_trace.csv
