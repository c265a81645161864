#!/bin/bash
# c3 (DCN: the head's two-segment, NI = 8 instantiation) old / new head kernel alternated on one box
set -u
O=gpurun_out/ab_head_c3.txt
: > $O
for r in 1 2 3 4; do
  for v in oldhead ""; do
    echo "# run $r head kernel: ${v:-new}" >> $O
    DCTR_LIB_VARIANT=$v DCTR_GEMM_MODE=split timeout 120 python tools/config_bench.py 1000 "c3 " 2>/dev/null | tail -n 1 | cut -c1-110 >> $O
  done
done
cat $O
