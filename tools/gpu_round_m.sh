#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_f2.py -q > $O/r03n_tests.log 2>&1; rc=$?
tail -6 $O/r03n_tests.log | cut -c1-300; grep -n "^E " $O/r03n_tests.log | cut -c1-300 | head -20
[ $rc -le 1 ] || exit $rc
rm -rf $O/prof_f2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f2 -- python3 tools/infer_bench.py 1 > $O/r03n_infer_prof.log 2>&1 || exit 2
cp $(ls $O/prof_f2/*/*kernel_stats.csv | head -1) $O/r03n_infer1_f2_kernel_stats.csv
cp $(ls $O/prof_f2/*/*kernel_trace.csv | head -1) $O/r03n_infer1_f2_kernel_trace.csv
rm -rf $O/prof_f2
head -12 $O/r03n_infer1_f2_kernel_stats.csv | cut -c1-160
