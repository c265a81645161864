#!/bin/bash
# Round-end run on the GPU box: both GPU suites (exact default, and the whole suite with the MLP products in split precision), the round's
# rocprofv3 summaries of `python bench.py`, the bench line (default and the driver's 20-step invocation), the stress loops, the other
# configs, the row-sharded path at world 1.  usage: bash tools/final_round_check.sh r05
set -u
TAG=${1:-r05}
O=gpurun_out/final_$TAG
mkdir -p $O
(timeout 700 python -m pytest tests -q -m gpu 2>&1 | tail -n 6) > $O/${TAG}_suite_exact_mode.txt
(DCTR_GEMM_MODE=split timeout 700 python -m pytest tests -q -m gpu 2>&1 | tail -n 6) > $O/${TAG}_suite_split_mode.txt
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
timeout 900 bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1
cp gpurun_out/prof_$TAG/${TAG}_*.txt $O/ 2>/dev/null
timeout 400 python bench.py > $O/${TAG}_bench.json 2> $O/bench.err
timeout 200 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_driver_invocation.json 2> $O/bench20.err
timeout 200 python bench.py --gemm-mode exact --no-cpu-baseline --no-end-to-end > $O/${TAG}_bench_exact_mode.json 2> $O/bench_exact.err
(timeout 300 python tools/default_path_stress.py 1000 exact; timeout 300 python tools/default_path_stress.py 1000 split) 2>&1 | grep -v amdgpu.ids > $O/${TAG}_default_path_stress.txt
(echo "# tools/config_bench.py 300, gemm_mode exact"; DCTR_GEMM_MODE=exact timeout 400 python tools/config_bench.py 300 2>/dev/null; echo "# gemm_mode split"; DCTR_GEMM_MODE=split timeout 400 python tools/config_bench.py 300 2>/dev/null) > $O/${TAG}_configs.txt
DCTR_FORCE_SHARDED=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-end-to-end > $O/${TAG}_bench_sharded_world1.json 2> $O/sharded.err
tail -n 3 $O/${TAG}_suite_exact_mode.txt $O/${TAG}_suite_split_mode.txt $O/${TAG}_default_path_stress.txt
