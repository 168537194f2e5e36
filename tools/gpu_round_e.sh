#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -q > $O/r03e_tests.log 2>&1; rc=$?
tail -8 $O/r03e_tests.log | cut -c1-300; grep -n "^E " $O/r03e_tests.log | cut -c1-300 | head -20
[ $rc -le 1 ] || exit $rc
timeout -k 10 300 python tools/kbench.py ctrgc > $O/r03e_kbench.log 2>&1; echo "kbench rc=$?"; grep "ctrgc_fwd\|dx3" $O/r03e_kbench.log | cut -c1-160
for sh in "64 64 64" "128 128 32" "256 256 16"; do timeout -k 10 200 python tools/ctrgc_phases.py $sh 2>&1 | grep -v amdgpu.ids | tee -a $O/r03e_phases.log; done
timeout -k 10 300 python tools/module_case_report.py "unit 64" > $O/r03e_modcases.log 2>&1; echo "modcases rc=$?"; grep -v amdgpu.ids $O/r03e_modcases.log | cut -c1-400
for f in sgd3s sgd3; do timeout -k 10 300 python tools/sgd_fixture_report.py $f > $O/r03e_$f.log 2>&1; echo "$f rc=$?"; done
grep -v "amdgpu.ids\|Warning\|warn\|losses.append" $O/r03e_sgd3s.log | cut -c1-250 | head -60
timeout -k 5 240 python -X faulthandler tools/stream_capture_check.py capture 4 2 > $O/r03e_stream42.log 2>&1; echo "stream 4/2 rc=$?"; grep -E "OK|differ|per model|replay" $O/r03e_stream42.log | cut -c1-300
rm -rf $O/prof_infer; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_infer -- python3 tools/infer_bench.py 1 > $O/r03e_infer_prof.log 2>&1; echo "infer prof rc=$?"
cp $(ls $O/prof_infer/*/*kernel_stats.csv | head -1) $O/r03e_infer1_kernel_stats.csv && rm -rf $O/prof_infer
head -30 $O/r03e_infer1_kernel_stats.csv | cut -c1-200
