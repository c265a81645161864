R=$PWD; cd /tmp; export TMPDIR=/tmp
for mm in 0 1e30; do
DCTR_AFM_TS_MIN_MACS=$mm rocprofv3 --kernel-trace --stats -d /tmp/pp_$mm -o afm -- python $R/tools/config_bench.py 100 "AFM run.sh:18 point B=128" > /tmp/log_$mm.txt 2>&1
grep ms_per /tmp/log_$mm.txt | cut -c60-130
(cd $R && python tools/prof_summary.py stats /tmp/pp_$mm/afm_results.db 2>&1 | head -16 | cut -c1-150; python tools/prof_summary.py timeline /tmp/pp_$mm/afm_results.db 2>&1 | head -24 | cut -c1-130)
done
