#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/r03aa_tests.log 2>&1; rc=$?
tail -3 $O/r03aa_tests.log | cut -c1-300; grep -n "^E " $O/r03aa_tests.log | cut -c1-300 | head -20
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/roofline_table.py > $O/r03aa_roofline_table.txt 2>&1; grep "KT5" $O/r03aa_roofline_table.txt | cut -c1-150
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/r03aa_bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/r03aa_bench.log | cut -c100-330
