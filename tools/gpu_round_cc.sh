#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_modules.py tests/test_gpu_blocks.py -q -x 2>&1 | tail -3 || exit 1
for h in 0 1; do echo "== TAMGCN_SPLIT_HALF=$h"; TAMGCN_SPLIT_HALF=$h timeout -k 10 300 python tools/conv_scaling.py 2>&1 | grep "split"; done
for h in 0 1 0 1; do TAMGCN_SPLIT_HALF=$h timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c100-200; done
