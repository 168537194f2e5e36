#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_modules.py tests/test_gpu_blocks.py -q -x > $O/r03ee_tests.log 2>&1; rc=$?
tail -3 $O/r03ee_tests.log | cut -c1-300; grep -n "^E " $O/r03ee_tests.log | cut -c1-300 | head -20
[ $rc -eq 0 ] || exit $rc
for h in 0 1; do echo "== TAMGCN_SPLIT_TALL=$h"; TAMGCN_SPLIT_TALL=$h timeout -k 10 300 python tools/conv_scaling.py 2>&1 | grep "split 768"; done
for h in 0 1 0 1; do TAMGCN_SPLIT_TALL=$h timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c100-200; done
