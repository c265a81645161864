#!/bin/bash
# c2: the lean step's optimizer placement (no output-layer detour on the grouping stream, one launch for every dense variable) at the large batch
R=$PWD; O=$R/gpurun_out; mkdir -p $O
res=$O/r06_knobs4.txt; : > $res
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-classic-reference --no-end-to-end"
run() { r=$(env "$@" timeout 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d.get('steady_ms_per_step'), d['host_enqueue_ms_per_step'])"); echo "$* : $r" >> $res; }
for rep in 1 2 3; do
run X=0
run DCTR_LEAN_BATCH=8192
done
cat $res
