#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/r03x_tests.log 2>&1; rc=$?
tail -4 $O/r03x_tests.log | cut -c1-300; grep -n "^E " $O/r03x_tests.log | cut -c1-300 | head -20
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --config 4stream --no-cpu-baseline > $O/r03x_b4.log 2>&1; echo "4stream rc=$?"; tail -1 $O/r03x_b4.log | cut -c1-330
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/r03x_bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/r03x_bench.log | cut -c1-330
