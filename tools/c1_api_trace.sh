#!/bin/bash
# c1 (B = 256, the reference's own operating point) is bound by the host's HIP calls: (a) exact / split mode alternated three times on ONE box
# (r05_configs.txt's 0.125 vs 0.168 on one box, 0.116 vs 0.118 on another: is the mode or the box the difference?), (b) the HIP API calls of
# 300 steps counted by rocprofv3 --hip-trace --stats (no counters in this pass)
set -u
O=gpurun_out/c1_api
mkdir -p $O
if [ "${SKIP_AB:-0}" != "1" ]; then
{
for r in 1 2 3; do
  for m in exact split; do
    echo "# run $r gemm_mode $m"
    DCTR_GEMM_MODE=$m timeout 120 python tools/config_bench.py 2000 "c1 " 2>/dev/null | tail -n 1
  done
done
} > $O/r05_c1_exact_vs_split.txt 2>&1
fi
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --hip-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o c1 -- python $GRAFT_REPO_ROOT/tools/config_bench.py 300 "c1 " > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/trace -name '*hip_api_stats.csv' | head -n 1)
if [ -n "$f" ]; then (echo "# rocprofv3 --hip-trace --stats -- python tools/config_bench.py 300 'c1 '   (10 warm-up + 300 timed steps + engine build)"; head -n 40 "$f") > $O/r05_c1_hip_api_stats.txt; fi
ls -la $(dirname "$f") > $O/trace_files.txt 2>&1
du -sh $O/trace >> $O/trace_files.txt 2>&1
rm -rf $O/trace
tail -n 5 $O/trace.log; head -n 30 $O/r05_c1_hip_api_stats.txt
