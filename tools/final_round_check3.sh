#!/bin/bash
# after a csrc change at the very end of the round: both suites once, fresh profile summaries, fresh bench lines (final_round_check2.sh minus the second exact-mode suite run)
set -u
TAG=${1:-r05}
O=gpurun_out/final_$TAG
mkdir -p $O
(timeout 700 python -m pytest tests -q -m gpu 2>&1 | tail -n 4) > $O/${TAG}_suite_exact_mode.txt
(DCTR_GEMM_MODE=split timeout 700 python -m pytest tests -q -m gpu 2>&1 | tail -n 4) > $O/${TAG}_suite_split_mode.txt

timeout 900 bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1
cp gpurun_out/prof_$TAG/${TAG}_*.txt $O/ 2>/dev/null
mkdir -p /tmp/pf && cp -r profiles /tmp/pf/ && cp gpurun_out/prof_$TAG/${TAG}_*.txt profiles/
timeout 400 python bench.py > $O/${TAG}_bench.json 2> $O/bench.err
timeout 200 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_driver_invocation.json 2> $O/bench20.err
tail -n 2 $O/${TAG}_suite_exact_mode.txt $O/${TAG}_suite_split_mode.txt
