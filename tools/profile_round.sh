#!/bin/bash
# GPU-box script: the round's profile artefacts.  usage: bash tools/profile_round.sh <tag> [commit]   (e.g. r04a $(git rev-parse --short HEAD))
#   gpurun_out/<tag>_bench_kernel_stats.csv          rocprofv3 --kernel-trace --stats of `python3 bench.py --single-mode`:
#                                                    the HEADLINE mode only (exact fp32 in every GEMM), so that a kernel's
#                                                    average duration in this file is the one bench.py's roofline.frac uses
#   gpurun_out/<tag>_bench_kernel_stats_split.csv    the same for `--gemm-mode split --single-mode` (the opt-in 2-term bf16 split)
#   gpurun_out/<tag>_bench_under_rocprof[_split].json  the bench lines of those runs
#   gpurun_out/<tag>_pmc_traffic.txt, traffic.json   FETCH_SIZE / WRITE_SIZE passes (separate runs) -> HBM bytes per launch
#   gpurun_out/<tag>_step_breakdown.txt              per-kernel ms/step with the side streams off (no overlap inflation), exact mode
#   gpurun_out/<tag>_bench_kernel_stats_serial.csv   rocprofv3 --stats of that one-stream run: kernels one at a time, as in bench.py's
#                                                    instrumented pass -- the file whose AverageNs matches roofline.avg_launch_us
#                                                    (in the default run the side streams put 2-3 kernels on the chip together and
#                                                    every one of them reads ~25 % longer)
#   gpurun_out/<tag>_bench.json                      the un-profiled default bench line (both modes, CPU baseline)
# Copy what should be judged into profiles/.
set -o pipefail
TAG=${1:-r01}
COMMIT=${2:-unknown}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
rm -rf $O/prof_stats $O/prof_stats_split $O/prof_fetch $O/prof_write $O/prof_serial
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --single-mode --no-cpu-baseline > $O/${TAG}_bench_under_rocprof.log 2>&1 || exit 2
grep "^{" $O/${TAG}_bench_under_rocprof.log | tail -1 > $O/${TAG}_bench_under_rocprof.json
cp $(ls $O/prof_stats/*/*kernel_stats.csv | head -1) $O/${TAG}_bench_kernel_stats.csv
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_split -- python3 bench.py --gemm-mode split --single-mode --no-cpu-baseline > $O/${TAG}_bench_under_rocprof_split.log 2>&1 || exit 2
grep "^{" $O/${TAG}_bench_under_rocprof_split.log | tail -1 > $O/${TAG}_bench_under_rocprof_split.json
cp $(ls $O/prof_stats_split/*/*kernel_stats.csv | head -1) $O/${TAG}_bench_kernel_stats_split.csv
echo "stats done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --single-mode > $O/pf.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --single-mode > $O/pw.log 2>&1 || exit 4
python tools/pmc_traffic.py $O/prof_fetch $O/prof_write $O/traffic.json "$TAG @ commit $COMMIT" > $O/${TAG}_pmc_traffic.txt || exit 5
echo "pmc done"
TAMGCN_SIDE_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --single-mode > $O/ps.log 2>&1 || exit 6
python tools/step_breakdown.py $(ls $O/prof_serial/*/*kernel_trace.csv | head -1) auto 60 > $O/${TAG}_step_breakdown.txt
cp $(ls $O/prof_serial/*/*kernel_stats.csv | head -1) $O/${TAG}_bench_kernel_stats_serial.csv
rm -rf $O/prof_stats $O/prof_stats_split $O/prof_fetch $O/prof_write $O/prof_serial
timeout -k 10 400 python bench.py > $O/${TAG}_bench.log 2>&1 || exit 7
grep "^{" $O/${TAG}_bench.log | tail -1 > $O/${TAG}_bench.json
head -3 $O/${TAG}_step_breakdown.txt
python tools/bench_summary.py $O/${TAG}_bench.json 3
