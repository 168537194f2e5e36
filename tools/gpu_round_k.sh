#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
{
echo "== fwd 64->64 T64 stats"; timeout -k 10 200 python tools/conv_phases.py --K 64 --M 64 --T 64 || exit 1
echo "== fwd 64->64 T64 nostats"; timeout -k 10 200 python tools/conv_phases.py --K 64 --M 64 --T 64 --nostats || exit 1
echo "== fwd 256->256 T16 stats"; timeout -k 10 200 python tools/conv_phases.py --K 256 --M 256 --T 16 || exit 1
echo "== bwd 64->64 T64 two"; timeout -k 10 200 python tools/conv_phases.py --K 64 --M 64 --T 64 --two --bwd || exit 1
echo "== bwd 192->64 T64"; timeout -k 10 200 python tools/conv_phases.py --K 192 --M 64 --T 64 --bwd || exit 1
} > $O/r03k_conv_phases.log 2>&1
grep -v "amdgpu.ids" $O/r03k_conv_phases.log | cut -c1-120
