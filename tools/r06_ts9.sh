#!/bin/bash
mkdir -p gpurun_out/ts9
python -m pytest tests -q -m gpu -k "afm or AFM or gemm_ts" > gpurun_out/ts9/afm_tests.txt 2>&1; tail -3 gpurun_out/ts9/afm_tests.txt
DCTR_GEMM_MODE=exact python -m pytest tests -q -m gpu -k "afm or AFM or gemm_ts" > gpurun_out/ts9/afm_tests_exact.txt 2>&1; tail -3 gpurun_out/ts9/afm_tests_exact.txt
for mode in split exact split exact; do
  echo "== DCTR_GEMM_MODE=$mode"
  DCTR_GEMM_MODE=$mode python tools/config_bench.py 200 "AFM reference point" 2>&1 | grep ms_per
done > gpurun_out/ts9/afm_step.txt 2>&1; cat gpurun_out/ts9/afm_step.txt
