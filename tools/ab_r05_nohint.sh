#!/bin/bash
# A/B on one box, split mode: the next batch's ids grouped ahead beside the table step (hint, default) vs grouped inside the step
B="python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-classic-reference --no-end-to-end"
run() { r=$(env "$@" timeout 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d.get('steady_ms_per_step'))"); echo "$* : $r (ms/step, steady)"; }
run X=default
run DCTR_BENCH_PREFETCH=0
run DCTR_BENCH_PREFETCH=0 DCTR_GROUP_AFTER=0
run DCTR_BENCH_PREFETCH=0 DCTR_GROUP_AFTER=2
run DCTR_BENCH_PREFETCH=0 DCTR_GROUP_AFTER=3
run X=default
