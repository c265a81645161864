#!/bin/bash
# A/B of the step's tail: fused scatter + table step (DCTR_FUSED_TAIL), where the prefetched grouping of the next batch starts
export DCTR_BENCH_TIMEOUT=100
run() { timeout 150 env "$@" python bench.py --steps 600 --warmup 50 --no-cpu-baseline 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$*', d['ms_per_step'], d['value'], {k:d['stage_ms'][k] for k in ('group_ids','scatter','tail')})" || tail -5 /tmp/err.txt; }
run DCTR_FUSED_TAIL=1
run DCTR_FUSED_TAIL=0
run DCTR_FUSED_TAIL=1
run DCTR_FUSED_TAIL=0
