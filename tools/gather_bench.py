"""Times the embedding gather in isolation (>=100 back-to-back launches) for a few table sizes / id distributions.
usage (GPU box): python tools/gather_bench.py [--K 16] [--B 4096]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from tf_repos_amd.engine import Engine, EngineConfig
from tf_repos_amd.synth import synth_batch


def run(V, K, B, F, uniform, iters):
    dev = torch.device("cuda", 0)
    eng = Engine(EngineConfig(model="deepfm", field_size=F, feature_size=V, embedding_size=K, deep_layers=(16,), dropout=(1.0,),
                              max_batch=B, table_mode="touched_rows"))
    ids, vals, labels = synth_batch(B, F, V, seed=7, uniform_ids=uniform)
    t = [torch.from_numpy(a).to(dev) for a in (ids, vals, labels)]
    eng.train_step(*t, want_loss=False)
    ms = min(eng.time_stage("embed_gather", iters=iters) for _ in range(3))
    nbytes = B * (F * (12 + 8 * K) + 8)
    eng.close()
    return {"V": V, "K": K, "B": B, "ids": "uniform" if uniform else "zipf", "table_MB": round(V * (K + 1) * 4 / 1e6, 1),
            "us": round(ms * 1e3, 3), "GBps": round(nbytes / ms / 1e6, 1)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--K", type=int, default=16)
    ap.add_argument("--B", type=int, default=4096)
    ap.add_argument("--F", type=int, default=39)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--big", action="store_true", help="also a table far larger than the 256 MB Infinity Cache")
    a = ap.parse_args()
    cases = [(1_000_000, False), (1_000_000, True)]
    if a.big:
        cases += [(64_000_000, True)]
    for V, uni in cases:
        print(json.dumps(run(V, a.K, a.B, a.F, uni, a.iters)), flush=True)
