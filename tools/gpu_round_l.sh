#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 300 python tools/f2_report.py > $O/r03l_f2_report.log 2>&1; grep "^l" $O/r03l_f2_report.log | cut -c1-120
timeout -k 10 600 python -m pytest tests/test_gpu_f2.py -q > $O/r03l_f2_tests.log 2>&1; rc=$?
tail -25 $O/r03l_f2_tests.log | cut -c1-250
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/infer_bench.py 1 2 4 8 16 32 > $O/r03l_infer_f2.log 2>&1; echo "infer rc=$?"; grep -v amdgpu.ids $O/r03l_infer_f2.log
TAMGCN_F2=0 timeout -k 10 300 python tools/infer_bench.py 1 2 4 8 16 32 64 > $O/r03l_infer_general.log 2>&1; echo "infer rc=$?"; grep -v amdgpu.ids $O/r03l_infer_general.log
timeout -k 10 300 python tools/infer_bench.py --t 52 1 16 > $O/r03l_infer_f2_t52.log 2>&1; grep -v amdgpu.ids $O/r03l_infer_f2_t52.log
