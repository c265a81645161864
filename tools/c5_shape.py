"""c5's table on ONE GPU (DeepFM vocab 1e8, K=32: 13.2 GB of parameters + 26.4 GB of Adam state; per-rank batch 8192): does the
engine size for the 288 GB part, and what do the HBM-bound kernels reach when nothing fits a cache?
usage (GPU box): python tools/c5_shape.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tf_repos_amd.engine import Engine, EngineConfig
from tf_repos_amd.synth import synth_batch

V, K, B, F = 100_000_000, 32, 8192, 39
out = {}
for mode in ("dense_exact", "touched_rows"):
    eng = Engine(EngineConfig(model="deepfm", field_size=F, feature_size=V, embedding_size=K, deep_layers=(400, 400, 400), dropout=(0.5, 0.5, 0.5),
                              l2_reg=1e-4, learning_rate=5e-4, optimizer="Adam", max_batch=B, seed=1, table_mode=mode))
    rng = np.random.default_rng(1)
    for pn, shp in eng.param_shapes.items():
        if pn not in ("emb", "linear"):                       # tables start at zero (a 12.8 GB host array is not the point here)
            eng.set_param(pn, rng.normal(0, 0.01, size=shp).astype(np.float32))
    ids, vals, labels = synth_batch(B, F, V, seed=5, uniform_ids=True)
    t = [torch.from_numpy(a).cuda() for a in (ids, vals, labels)]
    losses = [eng.train_step(*t) for _ in range(3)]
    assert all(np.isfinite(l) for l in losses), losses
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5 if mode == "dense_exact" else 50
    for _ in range(n):
        eng.train_step(*t, want_loss=False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    r = {"ms_per_step": round(ms, 3), "examples_per_sec": round(B / ms * 1e3, 1), "loss": [round(l, 5) for l in losses]}
    g = eng.time_stage("embed_gather", iters=20)
    r["gather_us"] = round(g * 1e3, 2)
    r["gather_GBps"] = round(B * (F * (12 + 8 * K) + 8) / g / 1e6, 1)
    if mode == "dense_exact":
        o = eng.time_stage("opt_table", iters=3)
        r["opt_table_ms"] = round(o, 3)
        r["opt_table_GBps"] = round((6 * V * (K + 1) * 4 + 4 * V) / o / 1e6, 1)
    r["mem_GB"] = round(torch.cuda.mem_get_info()[1] / 1e9 - torch.cuda.mem_get_info()[0] / 1e9, 1)
    out[mode] = r
    eng.close()
    torch.cuda.empty_cache()
print(json.dumps({"config": "c5 table on one GPU: DeepFM V=1e8 K=32 B=8192 uniform ids", **out}))
