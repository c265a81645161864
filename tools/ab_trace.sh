#!/bin/bash
# kernel-trace stats of bench.py c2 under two environments on the same box: in-step durations of the kernels matching a pattern
# usage: bash tools/ab_trace.sh <out name> <grep pattern> "<env A>" "<env B>"      -> gpurun_out/<out name>.txt
name=$1; pat=$2; shift 2
R=$PWD; OUT=$R/gpurun_out/prof_ab; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
res=$R/gpurun_out/$name.txt; : > $res
i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs DCTR_BENCH_TIMEOUT=200 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/t$i -o ab$i -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-classic-reference --no-end-to-end > $OUT/b$i.json 2> $OUT/t$i.err
  echo "== $envs: $(python -c "import json; d=json.loads(open('$OUT/b$i.json').readline()); print(d['ms_per_step'], 'ms/step under the tracer')")" >> $res
  (cd $R; python tools/prof_summary.py stats $OUT/t$i/ab${i}_results.db | grep -i "$pat\|calls" | head -12) >> $res
  rm -rf $OUT/t$i
done
cat $res
