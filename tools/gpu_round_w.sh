#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_modules.py tests/test_gpu_blocks.py -q -x 2>&1 | tail -3 || exit 1
python tools/kbench.py ctrgc 2>&1 | grep "bwd_de"
for i in 1 2 3 4 5 6; do
for spec in "capture 4 4" "capture 4 2"; do
    echo "=== $spec clips 128 T 64 (run $i)"; CHECK_CLIPS=128 CHECK_T=64 timeout -k 10 300 python -X faulthandler tools/stream_capture_check.py $spec 2>&1 | grep -v amdgpu | grep "replay\|differ\|per model\|by kind\|top of\|Error" | cut -c1-600
done; done > $O/r03w_capture_fixed.log 2>&1
grep -c "bit-identical" $O/r03w_capture_fixed.log; grep -c "differ" $O/r03w_capture_fixed.log; grep "differ\|top of" $O/r03w_capture_fixed.log | head -10
