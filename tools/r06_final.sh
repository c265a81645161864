#!/bin/bash
# round 6, last GPU action(s): both suites on the final library, fresh profile summaries (stamped with the sources' hash), fresh bench lines,
# the other configs in both modes, the c5 shard projection, the repeated-run stress of the default path.   usage: bash tools/r06_final.sh [part]
set -u
TAG=r06
O=gpurun_out/final_$TAG
mkdir -p $O
part=${1:-all}
if [ "$part" = "all" ] || [ "$part" = "suites" ]; then
(timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -n 6) > $O/${TAG}_suite_default_split_mode.txt
(DCTR_GEMM_MODE=exact timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -n 6) > $O/${TAG}_suite_exact_mode.txt
tail -n 2 $O/${TAG}_suite_default_split_mode.txt $O/${TAG}_suite_exact_mode.txt
fi
if [ "$part" = "all" ] || [ "$part" = "profile" ]; then
timeout 1200 bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1
cp gpurun_out/prof_$TAG/${TAG}_*.txt $O/ 2>/dev/null
cp gpurun_out/prof_$TAG/${TAG}_*.txt profiles/            # (bench.py reads the committed names: the box's copy for THIS run)
timeout 500 python bench.py > $O/${TAG}_bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_driver_invocation.json 2> $O/bench20.err
timeout 300 python bench.py --steps 20 --warmup 5 --gemm-mode exact --no-cpu-baseline --no-end-to-end > $O/${TAG}_bench_exact_mode.json 2> $O/bench_exact.err
fi
if [ "$part" = "all" ] || [ "$part" = "configs" ]; then
(echo "# tools/config_bench.py 300, default (split) mode"; timeout 500 python tools/config_bench.py 300 2>/dev/null; echo "# DCTR_GEMM_MODE=exact"; DCTR_GEMM_MODE=exact timeout 500 python tools/config_bench.py 300 2>/dev/null) > $O/${TAG}_configs.txt
TAG=$TAG timeout 600 bash tools/c5_shard_projection.sh > $O/c5.log 2>&1; cp gpurun_out/${TAG}_c5_shard_w1.json $O/ 2>/dev/null
(timeout 900 python tools/default_path_stress.py 300 split 2>&1 | tail -n 3) > $O/${TAG}_default_path_stress.txt
cat $O/${TAG}_configs.txt | cut -c1-140; tail -n 2 $O/${TAG}_default_path_stress.txt
fi
