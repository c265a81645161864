#!/bin/bash
# Kernel-trace stats of an arbitrary command on the GPU box: bash tools/prof_cmd.sh <tag> <limit_s> <command...>
# (every stage under `timeout`: a profiler that produced nothing must not leave a reader waiting on stdin)
set -u
TAG=$1; LIM=$2; shift 2
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# PMC="FETCH_SIZE" (one counter set per run, never together with other trace domains): mean counter value per dispatch instead
if [ -n "${PMC:-}" ]; then
    timeout $LIM rocprofv3 --kernel-trace --pmc $PMC -d $OUT/trace -o $TAG -- env -C $R "$@" > $OUT/cmd.out 2> $OUT/trace.err
else
    timeout $LIM rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG -- env -C $R "$@" > $OUT/cmd.out 2> $OUT/trace.err
fi
cd $R
if [ -f $OUT/trace/${TAG}_results.db ]; then
    if [ -n "${PMC:-}" ]; then
        timeout 120 python tools/prof_summary.py pmc $OUT/trace/${TAG}_results.db > $OUT/${TAG}_pmc.txt 2>&1
        head -25 $OUT/${TAG}_pmc.txt
        rm -rf $OUT/trace
        exit 0
    fi
    timeout 120 python tools/prof_summary.py stats $OUT/trace/${TAG}_results.db > $OUT/${TAG}_kernel_stats.txt 2>&1
    timeout 120 python tools/prof_summary.py timeline $OUT/trace/${TAG}_results.db > $OUT/${TAG}_timeline.txt 2>&1
    if [ -n "${TIMELINE:-}" ]; then head -60 $OUT/${TAG}_timeline.txt; else head -25 $OUT/${TAG}_kernel_stats.txt; fi
else
    echo "no results db"; tail -5 $OUT/trace.err
fi
rm -rf $OUT/trace
