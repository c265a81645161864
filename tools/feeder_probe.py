"""DeviceFeeder in isolation: synthetic numpy batches -> engine input slots -> train_step (debug / timing probe)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tf_repos_amd.engine import Engine, EngineConfig
from tf_repos_amd.feeder import DeviceFeeder
from tf_repos_amd.synth import synth_batch
B, F, V = 4096, 39, 1_000_000
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
eng = Engine(EngineConfig(model="deepfm", field_size=F, feature_size=V, embedding_size=16, deep_layers=(400, 400, 400), dropout=(0.5, 0.5, 0.5),
                          l2_reg=1e-4, learning_rate=5e-4, optimizer="Adam", max_batch=B, seed=1))
src = [synth_batch(B, F, V, seed=10 + i) for i in range(6)]
def gen():
    for s in range(steps):
        yield src[s % 6]
    yield tuple(a[:1000] for a in src[0])
for rep in range(2):
    fd = DeviceFeeder(eng, gen())
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    for ids, vals, labels, k in fd:
        eng.train_step(ids, vals, labels, want_loss=False)
        fd.release(k)
        n += int(labels.shape[0])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    fd.close()
    eng.check_ids()
    print("rep %d: %d examples in %.3f s = %.2f M examples/s" % (rep, n, dt, n / dt / 1e6), flush=True)
