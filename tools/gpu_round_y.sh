#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
for P in 0 -1 0 -1; do
TAMGCN_BENCH_MAIN_PRIORITY=$P timeout -k 10 300 python bench.py --no-cpu-baseline > $O/r03y_bench_m$P.log 2>&1; echo "main prio $P rc=$?"; tail -1 $O/r03y_bench_m$P.log | cut -c100-260
done
for P in 0 -1; do
TAMGCN_BENCH_MAIN_PRIORITY=$P timeout -k 10 300 python bench.py --config 4stream --no-cpu-baseline > $O/r03y_b4_m$P.log 2>&1; echo "4stream main prio $P rc=$?"; tail -1 $O/r03y_b4_m$P.log | cut -c100-260
done
