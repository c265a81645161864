#!/bin/bash
# counters of the sweep kernels in tools/lag_probe (alone on the chip)
R=$PWD; O=$R/gpurun_out; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU -d $O/lagpmc1 -o p -- $R/tools/_bin/lag_probe > /dev/null 2> $O/lagpmc1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_SCA SQ_WAVES -d $O/lagpmc2 -o p -- $R/tools/_bin/lag_probe > /dev/null 2> $O/lagpmc2.err
rocprofv3 --kernel-trace --stats -d $O/lagtr -o p -- $R/tools/_bin/lag_probe > /dev/null 2> $O/lagtr.err
cd $R
python tools/prof_summary.py pmc $O/lagpmc1/p_results.db $O/lagpmc2/p_results.db > $O/r06_lag_pmc.txt 2>&1
python tools/prof_summary.py stats $O/lagtr/p_results.db >> $O/r06_lag_pmc.txt 2>&1
tail -3 $O/lagpmc2.err >> $O/r06_lag_pmc.txt
rm -rf $O/lagpmc1 $O/lagpmc2 $O/lagtr
cat $O/r06_lag_pmc.txt | cut -c1-900
