#!/bin/bash
R=$PWD; O=$R/gpurun_out
res=$O/r06_ab_c1_lean2.txt; : > $res
run() { r=$(env "$@" timeout 100 python tools/config_bench.py 3000 "c1 " 2>/dev/null | tail -n 1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"); echo "$* : $r ms/step (c1, 3000 steps)" >> $res; }
for rep in 1 2 3; do
run X=0
run DCTR_WGRAD_SERIAL=1
run DCTR_CFG_HINT=0
run DCTR_WGRAD_SERIAL=1 DCTR_CFG_HINT=0
done
cat $res
