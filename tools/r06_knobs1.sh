#!/bin/bash
# round 6: schedule knobs re-measured on the round-6 library (same box, alternated)
R=$PWD; O=$R/gpurun_out; mkdir -p $O
res=$O/r06_knobs1.txt; : > $res
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-classic-reference --no-end-to-end"
run() { r=$(env "$@" timeout 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d.get('steady_ms_per_step'))"); echo "$* : $r" >> $res; }
for rep in 1 2; do
run X=0
run DCTR_WGRAD_LATE=1
run DCTR_WGRAD_LATE_LAYERS=1
run DCTR_WGRAD_LATE_LAYERS=2
run DCTR_WGRAD_LATE_LAYERS=3
run DCTR_WGRAD_SERIAL=1
run DCTR_DR3_SMALL=fdw
run DCTR_DR3_SMALL=fdw DCTR_WGRAD_LATE=1
run DCTR_DR3_SMALL=fdw DCTR_WGRAD_LATE_LAYERS=3
done
cat $res
