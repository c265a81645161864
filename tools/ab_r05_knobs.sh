#!/bin/bash
# A/B on one box, split mode (the round's default path): existing placement knobs re-measured now that the MLP products are shorter
B="python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-classic-reference --no-end-to-end"
run() { r=$(env "$@" timeout 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d.get('steady_ms_per_step'))"); echo "$* : $r (ms/step, steady)"; }
run X=default
run DCTR_GROUP_AFTER=2
run DCTR_GROUP_AFTER=1
run DCTR_LAG_BLOCKS_PER_CU=1
run DCTR_LAG_BLOCKS_PER_CU=4
run DCTR_SWEEP_PERIOD=4
run DCTR_SWEEP_PERIOD=16
run DCTR_PREGROUP_WAIT=none
run DCTR_SWEEP_AFTER_HEAD=1
run DCTR_OPT_SIDE=1
run X=default
