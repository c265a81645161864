#!/bin/bash
# the cheap rest of final_round_check.sh on the final library: smoke, exact-mode bench line, other configs in both modes, row-sharded path at world 1
set -u
TAG=${1:-r05}
O=gpurun_out/final_$TAG
mkdir -p $O
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
timeout 200 python bench.py --gemm-mode exact --no-cpu-baseline --no-end-to-end > $O/${TAG}_bench_exact_mode.json 2> $O/bench_exact.err
(echo "# tools/config_bench.py 300, gemm_mode exact"; DCTR_GEMM_MODE=exact timeout 300 python tools/config_bench.py 300 2>/dev/null; echo "# gemm_mode split"; DCTR_GEMM_MODE=split timeout 300 python tools/config_bench.py 300 2>/dev/null) > $O/${TAG}_configs.txt
DCTR_FORCE_SHARDED=1 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-end-to-end > $O/${TAG}_bench_sharded_world1.json 2> $O/sharded.err
tail -n 2 $O/smoke.txt; cut -c1-120 $O/${TAG}_configs.txt; python - <<'PY'
import json,sys
for f in ("r05_bench_exact_mode.json","r05_bench_sharded_world1.json"):
    try:
        d=json.loads(open("gpurun_out/final_r05/"+f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d.get("steady_ms_per_step"))
    except Exception as e: print(f, "ERR", e)
PY
