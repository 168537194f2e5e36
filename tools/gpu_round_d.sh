#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 300 python tools/module_case_report.py > $O/r03d_modcases.log 2>&1; echo "modcases rc=$?"; grep -v amdgpu.ids $O/r03d_modcases.log | cut -c1-400
: > $O/r03d_streams.log
for exp in "capture 2" "capture 3" "capture 4" "capture_noside 4"; do
  echo "=== $exp" >> $O/r03d_streams.log
  timeout -k 5 240 python -X faulthandler tools/stream_capture_check.py $exp >> $O/r03d_streams.log 2>&1; rc=$?
  echo "=== $exp rc=$rc" >> $O/r03d_streams.log
  [ $rc -ne 124 ] && [ $rc -ne 137 ] || exit $rc
done
grep -E "^===|OK|captured|differ|per model|Fatal|Error|replay" $O/r03d_streams.log | cut -c1-400 | head -60
timeout -k 10 300 python tools/kbench.py ctrgc > $O/r03d_kbench.log 2>&1; echo "kbench rc=$?"; grep "ctrgc_fwd\|dx3" $O/r03d_kbench.log | cut -c1-160
timeout -k 10 300 python tools/sgd_fixture_report.py sgd3s > $O/r03d_sgd3s.log 2>&1; echo "sgd rc=$?"; grep -v amdgpu.ids $O/r03d_sgd3s.log | cut -c1-220 | head -70
