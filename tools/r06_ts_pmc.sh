#!/bin/bash
# round 6: HBM traffic of the tall split-precision kernels in the AFM step (K = A = 256, B = 4096) by the PMC counters, separate passes
R=$PWD
mkdir -p gpurun_out/tspmc
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/ts_fetch -o afm -- python $R/tools/config_bench.py 10 "AFM reference point B=4096" > $R/gpurun_out/tspmc/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/ts_write -o afm -- python $R/tools/config_bench.py 10 "AFM reference point B=4096" > $R/gpurun_out/tspmc/write.log 2>&1
cd $R
python tools/prof_summary.py pmc /tmp/ts_fetch/afm_results.db /tmp/ts_write/afm_results.db > gpurun_out/tspmc/r06_afm_k256_pmc_traffic.txt 2>&1
head -40 gpurun_out/tspmc/r06_afm_k256_pmc_traffic.txt | cut -c1-220
