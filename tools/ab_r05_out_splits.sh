#!/bin/bash
# A/B: slabs of the output layer's gradient = blocks of the fused head kernel (128 today: 32 rows per block at c2, 64 at c4)
set -u
O=gpurun_out/ab_out_splits.txt
: > $O
for r in 1 2; do
for n in 128 256 512; do
  echo "# run $r DCTR_OUT_SPLITS=$n  c2 bench (split mode) 400 steps" >> $O
  DCTR_OUT_SPLITS=$n timeout 200 python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-classic-reference --no-end-to-end 2>>gpurun_out/ab_out_splits.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('ms_per_step','steady_ms_per_step')}, 'head alone', d.get('stage_ms',{}).get('head'))" >> $O
done
done
for n in 128 256 512; do
  echo "# DCTR_OUT_SPLITS=$n other configs (split mode)" >> $O
  for c in "c1 " "c3 " "c4 NFM" "c4 PNN-inner"; do
    DCTR_GEMM_MODE=split DCTR_OUT_SPLITS=$n timeout 120 python tools/config_bench.py 400 "$c" 2>/dev/null | tail -n 1 | cut -c1-110 >> $O
  done
done
cat $O
