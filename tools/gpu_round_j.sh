#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r03j_tests.log 2>&1; rc=$?
tail -6 $O/r03j_tests.log | cut -c1-300; grep -n "^E " $O/r03j_tests.log | cut -c1-300 | head -20
[ $rc -le 1 ] || exit $rc
timeout -k 10 300 python tools/kbench.py ctrgc > $O/r03j_kbench_ctrgc.log 2>&1; echo "kbench rc=$?"; grep "ctrgc" $O/r03j_kbench_ctrgc.log | cut -c1-150
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/r03j_bench_fork1.log 2>&1; echo "bench fork rc=$?"; tail -1 $O/r03j_bench_fork1.log | cut -c1-330
TAMGCN_BLOCK_FORK=0 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/r03j_bench_fork0.log 2>&1; echo "bench nofork rc=$?"; tail -1 $O/r03j_bench_fork0.log | cut -c1-330
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/r03j_bench_fork1b.log 2>&1; echo "bench fork rc=$?"; tail -1 $O/r03j_bench_fork1b.log | cut -c1-330
