#!/bin/bash
# c1 (B = 256): the lean small-batch step (one optimizer launch for every dense variable, no output-layer detour) vs round 5's placement
R=$PWD; O=$R/gpurun_out
res=$O/r06_ab_c1_lean.txt; : > $res
for rep in 1 2 3 4; do
  for lb in 0 512; do
    r=$(DCTR_LEAN_BATCH=$lb timeout 100 python tools/config_bench.py 3000 "c1 " 2>/dev/null | tail -n 1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "DCTR_LEAN_BATCH=$lb rep $rep: $r ms/step (c1, 3000 steps)" >> $res
  done
done
cat $res
