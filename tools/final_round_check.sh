set -u
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 bash tools/profile_round.sh r04 > gpurun_out/profile_round.log 2>&1; tail -3 gpurun_out/profile_round.log
mkdir -p profiles_new; cp gpurun_out/prof_r04/r04_*.txt profiles/ 2>/dev/null
timeout 600 python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err; tail -2 gpurun_out/r04_bench.err; cat gpurun_out/r04_bench.json | cut -c1-600
timeout 600 python tools/config_bench.py 100 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_configs.txt
timeout 400 bash tools/ab_shard.sh 2>&1 | tail -5
