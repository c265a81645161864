#!/bin/bash
# round 6, GPU call 1: the suite in the new default mode, the driver's bench invocation, the table-records A/B under the deferred join
# (round-5 verdict item 3).  usage (GPU box): bash tools/r06_run1.sh
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
(timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -n 15) > $O/r06a_suite_default.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $O/r06a_bench_driver.json 2> $O/r06a_bench_driver.err
res=$O/r06_ab_table_records.txt; : > $res
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-classic-reference --no-end-to-end"
for rep in 1 2 3; do
  for rec in 0 1; do
    r=$(DCTR_TABLE_RECORDS=$rec timeout 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d.get('steady_ms_per_step'))")
    echo "DCTR_TABLE_RECORDS=$rec rep $rep: $r (ms/step 200 steps, steady)" >> $res
  done
done
cd /tmp
for rec in 0 1; do
  DCTR_TABLE_RECORDS=$rec DCTR_BENCH_TIMEOUT=200 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_rec/t$rec -o rec$rec -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-classic-reference --no-end-to-end > $O/prof_rec_b$rec.json 2> $O/prof_rec_t$rec.err
  echo "== DCTR_TABLE_RECORDS=$rec under the tracer: $(python -c "import json,sys; d=json.loads(open('$O/prof_rec_b$rec.json').readline()); print(d['ms_per_step'], 'ms/step')")" >> $res
  (cd $R; python tools/prof_summary.py stats $O/prof_rec/t$rec/rec${rec}_results.db | grep -i "gather\|scatter_apply\|lag_advance\|group\|kernel  \|calls" | head -12) >> $res
  (cd $R; python tools/prof_summary.py timeline $O/prof_rec/t$rec/rec${rec}_results.db | head -30) >> $res
  rm -rf $O/prof_rec/t$rec
done
cd $R
echo "# sources sha256: $(python -c 'from tf_repos_amd.build import sources_hash; print(sources_hash())')" >> $res
cat $O/r06a_suite_default.txt | tail -5; head -c 600 $O/r06a_bench_driver.json; echo; cat $res | head -60
