#!/bin/bash
# c1 (B = 256): step variants, 6 alternated repetitions each, medians
R=$PWD; O=$R/gpurun_out
res=$O/r06_ab_c1_lean3.txt; : > $res
run() { r=$(env "$@" timeout 100 python tools/config_bench.py 3000 "c1 " 2>/dev/null | tail -n 1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"); echo "$* : $r" >> $res; }
for rep in 1 2 3 4 5 6; do
run DCTR_LEAN_BATCH=0 DCTR_GROUP_ONE_BLOCK=0
run DCTR_LEAN_BATCH=0
run DCTR_LEAN_WGRAD_LATE=0
run X=0
done
python - <<PY >> $res
import collections,statistics
d=collections.defaultdict(list)
for l in open("$res"):
    if " : " in l:
        k,v=l.rsplit(" : ",1); d[k].append(float(v))
for k,v in d.items(): print("median %-50s %.4f  (min %.4f max %.4f)" % (k, statistics.median(v), min(v), max(v)))
PY
tail -5 $res
