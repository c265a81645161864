#!/bin/bash
# kernel-trace stats + one-step timeline of the other BASELINE configs (tools/config_bench.py), for the round's notes
# usage (GPU box): bash tools/profile_configs.sh r04 "c3 DCN" "c4 PNN-inner" "c4 NFM"
TAG=$1; shift
R=$PWD; OUT=$R/gpurun_out/prof_cfg_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for name in "$@"; do
  key=$(echo "$name" | tr ' ' '_' | tr -cd 'A-Za-z0-9_-')
  DCTR_CFG_HINT=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/t_$key -o $key -- python $R/tools/config_bench.py 100 "$name" > $OUT/$key.json 2> $OUT/$key.err
  (cd $R; echo "== $name: $(tail -1 $OUT/$key.json)"; python tools/prof_summary.py stats $OUT/t_$key/${key}_results.db | head -22; python tools/prof_summary.py timeline $OUT/t_$key/${key}_results.db | head -45) > $OUT/${TAG}_$key.txt 2>&1
  rm -rf $OUT/t_$key
done
