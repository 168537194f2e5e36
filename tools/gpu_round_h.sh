#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r03h_tests.log 2>&1; rc=$?
tail -8 $O/r03h_tests.log | cut -c1-300; grep -n "^E " $O/r03h_tests.log | cut -c1-300 | head -20
[ $rc -le 1 ] || exit $rc
timeout -k 10 500 python bench.py --no-cpu-baseline > $O/r03h_bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/r03h_bench.log | cut -c1-400
