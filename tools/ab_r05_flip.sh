#!/bin/bash
# is the lag == classic comparison's occasional unit-sized difference the deferred join's doing?  the same test, 4 times each way
for dj in 0 1; do
  for rep in 1 2 3 4; do
    DCTR_DEFER_JOIN=$dj timeout 300 python -m pytest tests/test_bench_path_gpu.py -q -m gpu -s -k "lag_equals_classic_at_c2" 2>&1 | grep -E "mlp0/weights|emb  |passed|failed" | sed "s/^/defer_join=$dj rep $rep: /"
  done
done
