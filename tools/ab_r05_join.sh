#!/bin/bash
# A/B on one box: exact / split GEMM mode x deferred end-of-step join on / off (bench.py c2, 400 steps, the heavy extras off)
B="python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-classic-reference --no-end-to-end"
for mode in exact split; do
  for dj in 1 0; do
    for rep in 1 2; do
      r=$(DCTR_GEMM_MODE=$mode DCTR_DEFER_JOIN=$dj timeout 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'])")
      echo "gemm_mode=$mode defer_join=$dj rep $rep: $r ms/step"
    done
  done
done
