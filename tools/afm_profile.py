"""AFM step under rocprofv3 (kernel stats): python tools/afm_profile.py [model] [K] [V] [B] [layers]   (K = 256, V = 117581: the reference's run.sh point)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tf_repos_amd.engine import Engine, EngineConfig
from tf_repos_amd.synth import synth_batch
B, V, K = 4096, 1_000_000, 16
model = sys.argv[1] if len(sys.argv) > 1 else "afm"
if len(sys.argv) > 2: K = int(sys.argv[2])
if len(sys.argv) > 3: V = int(sys.argv[3])
if len(sys.argv) > 4: B = int(sys.argv[4])
layers = tuple(int(x) for x in sys.argv[5].split(',')) if len(sys.argv) > 5 else (400, 400, 400)
eng = Engine(EngineConfig(model=model, field_size=39, feature_size=V, embedding_size=K, deep_layers=layers if model != "afm" else (1,), dropout=(0.5,) * max(2, len(layers)),
                          attention_layers=(256,), l2_reg=1e-4, learning_rate=5e-4, optimizer="Adam", max_batch=B, seed=1))
rng = np.random.default_rng(1)
for pn, shp in eng.param_shapes.items():
    eng.set_param(pn, rng.normal(0, 0.01, size=shp).astype(np.float32))
ids, vals, labels = synth_batch(B, 39, V, seed=100)
t = [torch.from_numpy(a).cuda() for a in (ids, vals, labels)]
for s in range(12):
    eng.train_step(*t, want_loss=False)
torch.cuda.synchronize()
