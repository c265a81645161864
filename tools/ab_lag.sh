#!/bin/bash
# A/B of the time-blocked table sweep (csrc/lag.h): period 1 = classic
export DCTR_BENCH_TIMEOUT=100
run() { timeout 150 python bench.py --steps 600 --warmup 50 --no-cpu-baseline "$@" 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$*', d['ms_per_step'], d['value'], {k:d['stage_ms'][k] for k in ('embed_gather','tail')})" || tail -5 /tmp/err.txt; }
run --sweep-period 1
run --sweep-period 8
run --sweep-period 4
run --sweep-period 1
run --sweep-period 8
