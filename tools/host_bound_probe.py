"""Is a config's step host-bound?  Enqueue time of N steps (the loop returns before the GPU is done) against their wall time.
usage (GPU box): python tools/host_bound_probe.py [model] [K] [V] [B] [layers]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tf_repos_amd.engine import Engine, EngineConfig
from tf_repos_amd.synth import synth_batch
model = sys.argv[1] if len(sys.argv) > 1 else "deepfm"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
V = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
B = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
layers = tuple(int(x) for x in sys.argv[5].split(',')) if len(sys.argv) > 5 else (400, 400, 400)
eng = Engine(EngineConfig(model=model, field_size=39, feature_size=V, embedding_size=K, deep_layers=layers, dropout=(0.5,) * max(2, len(layers)),
                          attention_layers=(256,), l2_reg=1e-4, learning_rate=5e-4, optimizer="Adam", max_batch=B, seed=1))
rng = np.random.default_rng(1)
for pn, shp in eng.param_shapes.items():
    eng.set_param(pn, rng.normal(0, 0.01, size=shp).astype(np.float32))
batches = []
for i in range(4):
    ids, vals, labels = synth_batch(B, 39, V, seed=100 + i)
    si, sv, sl = eng.input_slot(i)
    si[:B].copy_(torch.from_numpy(ids)); sv[:B].copy_(torch.from_numpy(vals)); sl[:B].copy_(torch.from_numpy(labels))
    batches.append((si[:B], sv[:B], sl[:B]))
for s in range(20):
    eng.train_step(*batches[s % 4], want_loss=False)
torch.cuda.synchronize()
for N in (50, 200):
    t0 = time.perf_counter()
    for s in range(N):
        eng.train_step(*batches[s % 4], want_loss=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s K=%d B=%d: %d steps: enqueue %.1f us/step, wall %.1f us/step, GPU drained %.1f us after the loop" %
          (model, K, B, N, 1e6 * (t1 - t0) / N, 1e6 * (t2 - t0) / N, 1e6 * (t2 - t1)), flush=True)
