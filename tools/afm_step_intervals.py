"""Distribution of step-to-step intervals (GPU events on the engine's main stream) of AFM K = 256 at a small batch, un-traced.
usage: python tools/afm_step_intervals.py [B] [A] [steps]   (DCTR_AFM_TS_MIN_MACS=0 forces the tall kernels)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tf_repos_amd.engine import Engine, EngineConfig
from tf_repos_amd.synth import synth_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
A = int(sys.argv[2]) if len(sys.argv) > 2 else 128
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
V, K, F = 117581, 256, 39
eng = Engine(EngineConfig(model="afm", field_size=F, feature_size=V, embedding_size=K, deep_layers=(1,), dropout=(0.5, 0.5), attention_layers=(A,),
                          l2_reg=1e-4, learning_rate=5e-4, optimizer="Adam", max_batch=B, seed=1, use_graph=False))
rng = np.random.default_rng(1)
for pn, shp in eng.param_shapes.items():
    eng.set_param(pn, rng.normal(0, 0.01, size=shp).astype(np.float32))
batches = []
for i in range(4):
    ids, vals, labels = synth_batch(B, F, V, seed=100 + i)
    si, sv, sl = eng.input_slot(i)
    si[:B].copy_(torch.from_numpy(ids)); sv[:B].copy_(torch.from_numpy(vals)); sl[:B].copy_(torch.from_numpy(labels))
    batches.append((si[:B], sv[:B], sl[:B]))
import time
with torch.cuda.stream(eng.main_stream()):
    for s in range(20):
        eng.train_step(*batches[s % 4], want_loss=False); eng.prefetch_ids(batches[(s + 1) % 4][0])
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    host = []
    ev[0].record()
    t0 = time.perf_counter()
    for s in range(steps):
        h0 = time.perf_counter()
        eng.train_step(*batches[s % 4], want_loss=False); eng.prefetch_ids(batches[(s + 1) % 4][0])
        host.append(time.perf_counter() - h0)
        ev[s + 1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
iv = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(steps)])
host = np.array(host) * 1e3
print("B=%d A=%d: wall %.4f ms/step; GPU intervals ms: median %.4f mean %.4f p10 %.4f p90 %.4f max %.3f; host enqueue per step ms: median %.4f mean %.4f p90 %.4f" %
      (B, A, 1e3 * wall / steps, np.median(iv), iv.mean(), np.quantile(iv, 0.1), np.quantile(iv, 0.9), iv.max(), np.median(host), host.mean(), np.quantile(host, 0.9)))
