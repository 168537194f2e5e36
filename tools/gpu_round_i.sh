#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r03i_tests.log 2>&1; rc=$?
tail -6 $O/r03i_tests.log | cut -c1-300; grep -n "^E " $O/r03i_tests.log | cut -c1-300 | head -20
[ $rc -le 1 ] || exit $rc
timeout -k 10 500 python tools/ab_side_build.py TAMGCN_OLD_PROLOGUE 2 -- python tools/kbench.py conv > $O/r03i_ab_prologue.log 2>&1; echo "ab rc=$?"; grep "=====\|conv fwd 1x1\|bwd-data" $O/r03i_ab_prologue.log | cut -c1-130
timeout -k 10 300 python bench.py --config 4stream --no-cpu-baseline > $O/r03i_b4_seq.log 2>&1; echo "4stream seq rc=$?"; tail -1 $O/r03i_b4_seq.log | cut -c1-330
timeout -k 10 300 python bench.py --config 4stream --fork-streams 4 --no-graph --no-cpu-baseline > $O/r03i_b4_eager4.log 2>&1; echo "4stream eager4 rc=$?"; tail -1 $O/r03i_b4_eager4.log | cut -c1-330
timeout -k 10 300 python bench.py --config 4stream --no-graph --no-cpu-baseline > $O/r03i_b4_eager1.log 2>&1; echo "4stream eager1 rc=$?"; tail -1 $O/r03i_b4_eager1.log | cut -c1-330
timeout -k 10 400 python tools/config_bench.py ntu > $O/r03i_ntu.log 2>&1; echo "ntu rc=$?"; tail -1 $O/r03i_ntu.log | cut -c1-260
timeout -k 10 400 python tools/config_bench.py syn 128 > $O/r03i_syn128.log 2>&1; echo "syn rc=$?"; tail -1 $O/r03i_syn128.log | cut -c1-260
