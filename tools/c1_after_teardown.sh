#!/bin/bash
# r05_configs.txt reads c1 at 0.17 ms/step in its split-mode pass and 0.116 in its exact pass; alone, alternated, both modes read 0.114-0.117.
# Hypothesis: c1 is the FIRST config of each pass, and the split pass starts right behind the exit of the exact pass's process (its last
# engines hold GBs of device memory): the driver's teardown of that process disturbs the host-bound c1 loop.  Test: c1 right behind a
# heavy process's exit, in both modes, and again after a pause.
set -u
O=gpurun_out/r05_c1_after_teardown.txt
: > $O
for m in exact split; do
  DCTR_GEMM_MODE=exact timeout 200 python tools/config_bench.py 20 "AFM reference point B=4096" > /dev/null 2>&1
  echo "# c1, gemm_mode $m, started right behind the exit of a process that held the AFM K=256 B=4096 engine" >> $O
  DCTR_GEMM_MODE=$m timeout 100 python tools/config_bench.py 300 "c1 " 2>/dev/null | tail -n 1 >> $O
  sleep 8
  echo "# c1, gemm_mode $m, 8 s later" >> $O
  DCTR_GEMM_MODE=$m timeout 100 python tools/config_bench.py 300 "c1 " 2>/dev/null | tail -n 1 >> $O
done
cat $O
