#!/bin/bash
# Runs on the GPU box (gpurun): kernel-trace stats of bench.py + separate PMC passes for HBM traffic.
# usage: bash tools/profile_round.sh r01
set -u
TAG=${1:-r01}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
DCTR_BENCH_TIMEOUT=200 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-classic-reference --no-end-to-end ${BENCH_ARGS:-} > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
if [ "${TRACE_ONLY:-0}" != "1" ]; then     # (TRACE_ONLY=1: the kernel trace and its two summaries only; the counter summaries are kept)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o $TAG -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-classic-reference --no-end-to-end > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o $TAG -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-classic-reference --no-end-to-end > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc_sq -o $TAG -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-classic-reference --no-end-to-end > /dev/null 2> $OUT/pmc_sq.err
fi
cd $R
python tools/prof_summary.py stats $OUT/trace/${TAG}_results.db > $OUT/${TAG}_kernel_stats.txt 2>&1
python tools/prof_summary.py timeline $OUT/trace/${TAG}_results.db > $OUT/${TAG}_step_timeline.txt 2>&1
if [ "${TRACE_ONLY:-0}" != "1" ]; then
python tools/prof_summary.py pmc $OUT/pmc_fetch/${TAG}_results.db $OUT/pmc_write/${TAG}_results.db > $OUT/${TAG}_pmc_traffic.txt 2>&1
python tools/prof_summary.py pmc $OUT/pmc_sq/${TAG}_results.db > $OUT/${TAG}_pmc_sq.txt 2>&1
fi
# stamp: which sources these summaries describe (bench.py: profile_is_stale)
stamp="# sources sha256: $(python -c 'from tf_repos_amd.build import sources_hash; print(sources_hash())')"
for f in $OUT/${TAG}_kernel_stats.txt $OUT/${TAG}_step_timeline.txt $OUT/${TAG}_pmc_traffic.txt $OUT/${TAG}_pmc_sq.txt; do
  [ -f $f ] && ! grep -q "^# sources sha256" $f && echo "$stamp" >> $f
done
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
cat $OUT/${TAG}_step_timeline.txt | head -60
