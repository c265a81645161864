#!/bin/bash
# A/B on one box: MLP optimizer launches at the tail of the weight-gradient stream (one launch + one re-split) vs per layer between the weight gradients
B="python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-classic-reference --no-end-to-end"
for mode in split exact; do
  for ot in 1 0; do
    for rep in 1 2; do
      r=$(DCTR_OPT_TAIL=$ot timeout 200 $B --gemm-mode $mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d.get('steady_ms_per_step'), d.get('final_flush_ms'))")
      echo "gemm_mode=$mode opt_tail=$ot rep $rep: $r (ms/step, steady, final flush ms)"
    done
  done
done
