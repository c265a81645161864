#!/bin/bash
# kernel-trace stats of bench.py c2 under the two table layouts (DCTR_TABLE_RECORDS=0 / 1), same box: the in-step durations of the
# kernels that touch table rows.   usage: bash tools/ab_records_trace.sh   -> gpurun_out/r04_ab_records_kernels.txt
R=$PWD; OUT=$R/gpurun_out/prof_rec; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
res=$R/gpurun_out/r04_ab_records_kernels.txt; : > $res
for rec in 0 1; do
  DCTR_TABLE_RECORDS=$rec DCTR_BENCH_TIMEOUT=200 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/t$rec -o rec$rec -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-classic-reference --no-end-to-end > $OUT/b$rec.json 2> $OUT/t$rec.err
  echo "== DCTR_TABLE_RECORDS=$rec: $(python -c "import json,sys; d=json.loads(open('$OUT/b$rec.json').readline()); print(d['ms_per_step'], 'ms/step under the tracer')")" >> $res
  (cd $R; python tools/prof_summary.py stats $OUT/t$rec/rec${rec}_results.db | grep -i "gather\|scatter_apply\|lag_advance\|group\|kernel  \|calls" | head -12) >> $res
  rm -rf $OUT/t$rec
done
cat $res
