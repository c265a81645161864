"""K=32 gather lane mappings on an HBM-resident table (V=1e8, uniform ids, B=8192): DCTR_GATHER_K32=0|1|2"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tf_repos_amd.engine import Engine, EngineConfig
from tf_repos_amd.synth import synth_batch
V, K, B, F = 100_000_000, 32, 8192, 39
eng = Engine(EngineConfig(model="deepfm", field_size=F, feature_size=V, embedding_size=K, deep_layers=(400, 400, 400), dropout=(0.5, 0.5, 0.5),
                          l2_reg=1e-4, learning_rate=5e-4, optimizer="Adam", max_batch=B, seed=1, table_mode="touched_rows"))
ids, vals, labels = synth_batch(B, F, V, seed=5, uniform_ids=True)
t = [torch.from_numpy(a).cuda() for a in (ids, vals, labels)]
eng.train_step(*t)
for _ in range(3):
    g = eng.time_stage("embed_gather", iters=30)
    print(os.environ.get("DCTR_GATHER_K32", "0"), round(g * 1e3, 2), "us", round(B * (F * (12 + 8 * K) + 8) / g / 1e6, 1), "GB/s", flush=True)
