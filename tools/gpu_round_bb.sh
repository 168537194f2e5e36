#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for mt in 0 2 1; do echo "== TAMGCN_KX1_MT=$mt"; TAMGCN_KX1_MT=$mt timeout -k 10 300 python tools/tconv_scaling.py 2>&1 | grep -v amdgpu; done
for mt in 0 2 1 0 2; do TAMGCN_KX1_MT=$mt timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c100-200; done
