#!/bin/bash
B="python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-classic-reference --no-end-to-end"
for sm in fd fdw ""; do
  for rep in 1 2; do
    r=$(DCTR_DR3_SMALL=$sm timeout 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d.get('steady_ms_per_step'), d['roofline']['family']['us_per_step_sum_of_dispatches'])")
    echo "DCTR_DR3_SMALL='$sm' rep $rep: $r (ms/step, steady, sum of the nine dispatches us)"
  done
done
