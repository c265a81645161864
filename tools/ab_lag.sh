#!/bin/bash
# A/B runs of the c2 step: knobs as environment variables
export DCTR_BENCH_TIMEOUT=100
run() { timeout 150 env "$@" python bench.py --steps 600 --warmup 50 --no-cpu-baseline --no-classic-reference 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$*', d['ms_per_step'], d['value'], d['roofline']['ms'])" || tail -5 /tmp/err.txt; }
run A=0
run A=0
run DCTR_SWEEP_PERIOD=4
