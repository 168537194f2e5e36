#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 300 python tools/eval_bwd_report.py ucla_t13 > $O/r03g_evalbwd.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/r03g_evalbwd.log | cut -c1-300
TAMGCN_CTRGC_CT=8 timeout -k 10 300 python tools/eval_bwd_report.py ucla_t13 > $O/r03g_evalbwd_ct8.log 2>&1; echo "ct8 rc=$?"; grep -v amdgpu.ids $O/r03g_evalbwd_ct8.log | head -8 | cut -c1-300
timeout -k 10 300 python -m pytest tests/test_gpu_stgcn.py -m gpu -q 2>&1 | tail -5 | cut -c1-300
