#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -q > $O/r03f_tests.log 2>&1; rc=$?
tail -8 $O/r03f_tests.log | cut -c1-300; grep -n "^E " $O/r03f_tests.log | cut -c1-300 | head -20
[ $rc -le 1 ] || exit $rc
for ct in 16 8; do TAMGCN_CTRGC_CT=$ct TAMGCN_DEBUG_OCC=1 timeout -k 10 300 python tools/kbench.py ctrgc > $O/r03f_kbench_ct$ct.log 2>&1; echo "kbench ct=$ct rc=$?"; grep "ctrgc_fwd (E loaded, x3\|dx3" $O/r03f_kbench_ct$ct.log | cut -c1-150; done
grep "^\[tamgcn\]" $O/r03f_kbench_ct16.log | sort | uniq -c
for sh in "64 64 64" "256 256 16"; do timeout -k 10 200 python tools/ctrgc_phases.py $sh 2>&1 | grep -v amdgpu.ids | tee -a $O/r03f_phases.log; done
timeout -k 10 500 python bench.py --no-cpu-baseline > $O/r03f_bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/r03f_bench.log | cut -c1-700
