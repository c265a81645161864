#!/bin/bash
# what the in-step timer's sampled step costs a 20-step timed region (the driver's invocation)
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-classic-reference --no-end-to-end"
for t in 2 0 2 0 2 0; do
  r=$(DCTR_BENCH_TIMER=$t timeout 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d.get('steady_ms_per_step'), d.get('final_flush_ms'))")
  echo "DCTR_BENCH_TIMER=$t (2: every product of steps 3, 35, .. carries events; 0: off): $r (ms/step of the 20 steps, steady, fixed part ms)"
done
