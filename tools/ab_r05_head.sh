#!/bin/bash
# A/B on ONE box: the fused head kernel with a block's loads all in flight (product library) against the round's earlier kernel
# (python -m tf_repos_amd.build --variant oldhead from the previous commit's dense_ops.hip; DCTR_LIB_VARIANT=oldhead)
set -u
O=gpurun_out/ab_head.txt
: > $O
for r in 1 2 3; do
  for v in oldhead ""; do
  echo "# run $r c2 bench (split mode) 400 steps, head kernel: ${v:-new}" >> $O
  DCTR_LIB_VARIANT=$v timeout 200 python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-classic-reference --no-end-to-end 2>>gpurun_out/ab_head.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('ms_per_step','steady_ms_per_step')}, 'head alone', d.get('stage_ms',{}).get('head'))" >> $O
  done
done
for v in oldhead ""; do
echo "# other configs (split mode), head kernel: ${v:-new}" >> $O
for c in "c3 " "c4 NFM" "c4 PNN-inner"; do
  DCTR_LIB_VARIANT=$v DCTR_GEMM_MODE=split timeout 120 python tools/config_bench.py 400 "$c" 2>/dev/null | tail -n 1 | cut -c1-110 >> $O
done
done
cat $O
