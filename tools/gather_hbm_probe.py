"""The gather kernel alone on HBM-resident tables (uniform ids, far beyond the 256 MB Infinity Cache), K = 16 and K = 32: time per
launch and -- under `PMC=FETCH_SIZE bash tools/prof_cmd.sh ...` -- the bytes the memory side actually moved per launch.
usage (GPU box): python tools/gather_hbm_probe.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

dev = torch.device("cuda", 0)
for K, V in ((16, 64 * 1024 * 1024), (32, 32 * 1024 * 1024)):
    ms, alg = bench.hbm_resident_gather(dev, K=K, V=V, iters=int(os.environ.get("ITERS", "200")))
    print(json.dumps({"K": K, "V": V, "table_GB": round(V * (K + 1) * 4 / 1e9, 2), "us_per_launch": round(1e3 * ms, 2), "algorithmic_MB": round(alg / 1e6, 2),
                      "algorithmic_TBps": round(alg / ms / 1e9, 3)}), flush=True)
