#!/bin/bash
R=$PWD; O=$R/gpurun_out; mkdir -p $O
res=$O/r06_knobs5.txt; : > $res
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-classic-reference --no-end-to-end"
run() { r=$(env "$@" timeout 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d.get('steady_ms_per_step'), d['host_enqueue_ms_per_step'])"); echo "$* : $r" >> $res; }
for rep in 1 2 3; do
run X=0
run DCTR_PREGROUP_WAIT=end
run DCTR_PREGROUP_WAIT=none
done
cat $res
