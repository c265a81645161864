#!/bin/bash
# round 6: AFM at the reference's batch (B = 128, K = A = 256) under rocprofv3, split (tall bf16 products) against exact
R=$PWD
mkdir -p gpurun_out/ts2
cd /tmp && export TMPDIR=/tmp
for mode in split exact; do
  DCTR_GEMM_MODE=$mode rocprofv3 --kernel-trace --stats -d /tmp/prof_$mode -o afm -- python $R/tools/config_bench.py 100 "AFM reference point B=${AFM_B:-128}" > $R/gpurun_out/ts2/b${AFM_B:-128}_$mode.log 2>&1
  grep ms_per_step $R/gpurun_out/ts2/b${AFM_B:-128}_$mode.log
  (cd $R && python tools/prof_summary.py stats /tmp/prof_$mode/afm_results.db > gpurun_out/ts2/b${AFM_B:-128}_${mode}_stats.txt 2>&1; python tools/prof_summary.py timeline /tmp/prof_$mode/afm_results.db > gpurun_out/ts2/b${AFM_B:-128}_${mode}_timeline.txt 2>&1)
  head -30 $R/gpurun_out/ts2/b${AFM_B:-128}_${mode}_stats.txt | cut -c1-200
done
