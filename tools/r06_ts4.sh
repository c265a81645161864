#!/bin/bash
# where the tall split-precision products start to pay in the AFM step: B = 128 .. 4096 at K = A = 256, with (min rows 0) and without (huge)
mkdir -p gpurun_out/ts4
for minrows in 0 1000000000; do
  echo "== DCTR_AFM_TS_MIN_ROWS=$minrows"
  DCTR_AFM_TS_MIN_ROWS=$minrows python tools/config_bench.py 100 "AFM reference point" 2>&1 | grep ms_per
done > gpurun_out/ts4/afm_step.txt 2>&1; cat gpurun_out/ts4/afm_step.txt
