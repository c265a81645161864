"""Times individual stages of the c2 step (engine.time_stage) -- for A/B experiments with env-var knobs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tf_repos_amd.engine import Engine, EngineConfig
from tf_repos_amd.synth import synth_batch
stages = sys.argv[1:] or ["opt_table"]
dev = torch.device("cuda:0")
eng = Engine(EngineConfig(model="deepfm", field_size=39, feature_size=1_000_000, embedding_size=16, deep_layers=(400, 400, 400),
                          dropout=(0.5, 0.5, 0.5), l2_reg=1e-4, learning_rate=5e-4, optimizer="Adam", max_batch=4096, seed=1))
rng = np.random.default_rng(1)
for name, shp in eng.param_shapes.items():
    eng.set_param(name, rng.normal(0, 0.01, size=shp).astype(np.float32))
ids, vals, labels = synth_batch(4096, 39, 1_000_000, seed=5)
t = lambda a: torch.from_numpy(a).to(dev)
for _ in range(3):
    eng.train_step(t(ids), t(vals), t(labels), want_loss=False)
for s in stages:
    print(s, " ".join("%.1f" % (1e3 * eng.time_stage(s, iters=30)) for _ in range(3)), "us")
