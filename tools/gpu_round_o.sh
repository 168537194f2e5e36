#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r03p_tests.log 2>&1; rc=$?
tail -4 $O/r03p_tests.log | cut -c1-300; grep -n "^E " $O/r03p_tests.log | cut -c1-300 | head -20
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/kbench.py ctrgc > $O/r03p_kbench_ctrgc.log 2>&1; echo "kbench rc=$?"; grep "bwd_de" $O/r03p_kbench_ctrgc.log | cut -c1-130
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/r03p_bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/r03p_bench.log | cut -c1-330
