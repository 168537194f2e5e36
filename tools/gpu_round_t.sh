#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/r03t_tests.log 2>&1; rc=$?
tail -4 $O/r03t_tests.log | cut -c1-300; grep -n "^E " $O/r03t_tests.log | cut -c1-300 | head -20
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/kbench.py conv > $O/r03t_kbench_conv.log 2>&1; echo "kbench rc=$?"; grep "bwd-data" $O/r03t_kbench_conv.log | cut -c1-130
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/r03t_bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/r03t_bench.log | cut -c1-330
