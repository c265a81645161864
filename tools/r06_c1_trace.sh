#!/bin/bash
# c1 (B = 256): step time alternated with the kernel timeline of one step under the tracer
R=$PWD; O=$R/gpurun_out; export TMPDIR=/tmp
for i in 1 2 3; do timeout 100 python tools/config_bench.py 2000 "c1 " 2>/dev/null | tail -n 1; done > $O/r06_c1.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/c1tr -o c1 -- python $R/tools/config_bench.py 600 "c1 " > /dev/null 2> $O/c1tr.err
cd $R
python tools/prof_summary.py timeline $O/c1tr/c1_results.db >> $O/r06_c1.txt 2>&1
python tools/prof_summary.py stats $O/c1tr/c1_results.db | head -30 >> $O/r06_c1.txt 2>&1
rm -rf $O/c1tr
cat $O/r06_c1.txt
