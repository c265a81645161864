#!/bin/bash
# GPU box: single-GPU bench under a few env settings; prints ms_per_step per setting
for cfg in "$@"; do
  env $cfg timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$cfg', d['ms_per_step'], 'ms/step', d['value'], 'ex/s')
"
done
