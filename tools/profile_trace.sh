#!/bin/bash
# Runs on the GPU box (gpurun): kernel-trace stats + one-step timeline of bench.py only (no PMC passes).
# usage: [BENCH_ARGS=..] bash tools/profile_trace.sh tag
set -u
TAG=${1:-trace}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
DCTR_BENCH_TIMEOUT=200 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-classic-reference --no-end-to-end ${BENCH_ARGS:-} > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
cd $R
python tools/prof_summary.py stats $OUT/trace/${TAG}_results.db > $OUT/${TAG}_kernel_stats.txt 2>&1
python tools/prof_summary.py timeline $OUT/trace/${TAG}_results.db > $OUT/${TAG}_step_timeline.txt 2>&1
rm -rf $OUT/trace
