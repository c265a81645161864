#!/bin/bash
# the sharded step forced onto one GPU (RCCL at world 1): owner-side lag on / off, and the selftest
export DCTR_BENCH_TIMEOUT=150 DCTR_FORCE_SHARDED=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
run() { timeout 200 env "$@" python bench.py --gpus 1 --steps 400 --warmup 40 --no-cpu-baseline 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$*', d.get('ms_per_step'), d.get('value'), d.get('selftest'), d.get('loss_n_ranks'), d.get('loss_one_rank'))" || tail -8 /tmp/err.txt; }
run A=0
run DCTR_OWNER_LAG=0
run A=0
timeout 200 python bench.py --gpus 1 --selftest 2>/tmp/err.txt || tail -8 /tmp/err.txt
