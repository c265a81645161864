#!/bin/bash
# round 6: the three tall split-precision products in the AFM step -- op tests, AFM parity tests, step time by batch, kernel stats at B = 4096
R=$PWD
mkdir -p gpurun_out/ts12
timeout 300 tools/_bin/gemm_ts_probe > gpurun_out/ts12/probe.txt 2>&1; grep -v "max .err" gpurun_out/ts12/probe.txt | tail -14
python -m pytest tests/test_gemm_ts_gpu.py -q -m gpu -s > gpurun_out/ts12/ops.txt 2>&1; grep -E "passed|failed|Error" gpurun_out/ts12/ops.txt | head
python -m pytest tests -q -m gpu -k "afm or AFM" > gpurun_out/ts12/afm_tests.txt 2>&1; tail -3 gpurun_out/ts12/afm_tests.txt
for mode in split exact; do
  echo "== DCTR_GEMM_MODE=$mode"
  DCTR_GEMM_MODE=$mode python tools/config_bench.py 100 "AFM reference point" 2>&1 | grep ms_per
done > gpurun_out/ts12/afm_step.txt 2>&1; cat gpurun_out/ts12/afm_step.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_afm -o afm -- python $R/tools/config_bench.py 100 "AFM reference point B=4096" > $R/gpurun_out/ts12/b4096_trace.log 2>&1
cd $R && python tools/prof_summary.py stats /tmp/prof_afm/afm_results.db > gpurun_out/ts12/b4096_stats.txt 2>&1; python tools/prof_summary.py timeline /tmp/prof_afm/afm_results.db > gpurun_out/ts12/b4096_timeline.txt 2>&1
head -22 gpurun_out/ts12/b4096_stats.txt | cut -c1-170
