#!/bin/bash
# round 6: the sweep's variants in the step (same box, alternated) + the lagging-rows tests on the new replay loop
R=$PWD; O=$R/gpurun_out; mkdir -p $O
(timeout 900 python -m pytest tests/test_lag_gpu.py tests/test_bench_path_gpu.py tests/test_ieee_adam_gpu.py -q -m gpu -x 2>&1 | tail -n 8) > $O/r06b_lag_tests.txt
res=$O/r06_knobs2.txt; : > $res
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-classic-reference --no-end-to-end"
run() { r=$(env "$@" timeout 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d.get('steady_ms_per_step'))"); echo "$* : $r" >> $res; }
for rep in 1 2 3; do
run X=0
run DCTR_LAG_SMALL=1
run DCTR_LAG_PIPE=1
run DCTR_LAG_PIPE=1 DCTR_LAG_PIPE_BLOCKS_PER_CU=2
run DCTR_LAG_BLOCKS_PER_CU=4
done
cat $O/r06b_lag_tests.txt; cat $res
