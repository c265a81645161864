#!/bin/bash
# where the tall split-precision products start to pay in the AFM step: B = 128 .. 4096 at K = A = 256, with (min rows 0) and without (huge)
mkdir -p gpurun_out/ts4
for rep in 1 2; do
for minrows in 0 262144; do
  echo "== DCTR_AFM_TS_MIN_ROWS=$minrows"
  DCTR_AFM_TS_MIN_ROWS=$minrows python tools/config_bench.py 300 "AFM reference point B=${AFM_B:-128}" 2>&1 | grep ms_per
done
done > gpurun_out/ts4/afm_step_b${AFM_B:-128}.txt 2>&1; cat gpurun_out/ts4/afm_step_b${AFM_B:-128}.txt
