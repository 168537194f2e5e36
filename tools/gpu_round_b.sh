#!/bin/bash
# GPU-box script (round 3, call B): the -m gpu suite on the CT = 8 fused CTRGC kernels, the multi-stream capture bisect
# (every experiment in its own process), the bench line.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -q > $O/r03b_tests.log 2>&1; rc=$?
tail -40 $O/r03b_tests.log | cut -c1-400
[ $rc -le 1 ] || exit $rc
: > $O/r03b_streams.log
for exp in "toy 2" "capture_fwd 2" "capture_one 1" "capture_noside 2" "capture 2" "capture_relaxed 2" "capture 4"; do
  echo "=== $exp" >> $O/r03b_streams.log
  timeout -k 5 180 python -X faulthandler tools/stream_capture_check.py $exp >> $O/r03b_streams.log 2>&1; rc=$?
  echo "=== $exp rc=$rc" >> $O/r03b_streams.log
  [ $rc -ne 124 ] && [ $rc -ne 137 ] || exit $rc
done
grep -E "^===|OK|captured|Fatal|File \"/root|Error" $O/r03b_streams.log | cut -c1-300 | head -60
timeout -k 10 500 python bench.py > $O/r03b_bench.log 2>&1; rc=$?
echo "bench rc=$rc"; tail -1 $O/r03b_bench.log | cut -c1-3000
