#!/bin/bash
# GPU-box script (round 3, call C): -m gpu suite, CTRGC kernel timings with the occupancy the runtime reports, configs[2]
# on four model streams and in order, batch-1 inference kernel list.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -q > $O/r03c_tests.log 2>&1; rc=$?
tail -12 $O/r03c_tests.log | cut -c1-300
[ $rc -le 1 ] || exit $rc
TAMGCN_DEBUG_OCC=1 timeout -k 10 300 python tools/kbench.py ctrgc > $O/r03c_kbench.log 2>&1; rc=$?
echo "kbench rc=$rc"; grep -v "^\[tamgcn\]" $O/r03c_kbench.log | tail -40 | cut -c1-200; grep "^\[tamgcn\]" $O/r03c_kbench.log | sort | uniq -c | head
[ $rc -ne 124 ] && [ $rc -ne 137 ] || exit $rc
for f in 1 0; do
  timeout -k 10 300 python -X faulthandler bench.py --config 4stream --fork-streams $f --no-cpu-baseline > $O/r03c_b4_fork$f.log 2>&1; rc=$?
  echo "4stream fork=$f rc=$rc"; tail -1 $O/r03c_b4_fork$f.log | cut -c1-400
  [ $rc -ne 124 ] && [ $rc -ne 137 ] || exit $rc
done
timeout -k 10 200 python tools/infer_kernels.py 1 > $O/r03c_infer1.log 2>&1; echo "infer rc=$?"; head -40 $O/r03c_infer1.log | cut -c1-160
