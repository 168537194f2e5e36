#!/bin/bash
# GPU-box script (round 3, call A): the -m gpu suite, then configs[2] with the four models on four streams (eager warm-up and
# HIP-graph capture; python -X faulthandler so that an abort leaves a stack), the sequential form, and the bench line.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q > $O/r03a_tests.log 2>&1; rc=$?
tail -15 $O/r03a_tests.log
[ $rc -le 1 ] || exit $rc
timeout -k 10 300 python -X faulthandler bench.py --config 4stream --fork-streams 1 --no-cpu-baseline > $O/r03a_b4fork.log 2>&1; rc=$?
echo "4stream fork rc=$rc"; tail -4 $O/r03a_b4fork.log | cut -c1-600
[ $rc -ne 124 ] && [ $rc -ne 137 ] || exit $rc
timeout -k 10 300 python bench.py --config 4stream --fork-streams 0 --no-cpu-baseline > $O/r03a_b4seq.log 2>&1; rc=$?
echo "4stream seq rc=$rc"; tail -2 $O/r03a_b4seq.log | cut -c1-600
[ $rc -ne 124 ] && [ $rc -ne 137 ] || exit $rc
timeout -k 10 500 python bench.py > $O/r03a_bench.log 2>&1; rc=$?
echo "bench rc=$rc"; tail -1 $O/r03a_bench.log | cut -c1-1500
