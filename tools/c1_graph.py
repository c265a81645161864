import sys, time, os
sys.path.insert(0, '.')
import numpy as np, torch
from tf_repos_amd.engine import Engine, EngineConfig
from tf_repos_amd.synth import synth_batch
B, V, K = 256, 117581, 8
for ug in (False, True):
    eng = Engine(EngineConfig(model="deepfm", field_size=39, feature_size=V, embedding_size=K, deep_layers=(400,400,400), dropout=(0.5,0.5,0.5),
                              l2_reg=1e-4, learning_rate=5e-4, optimizer="Adam", max_batch=B, seed=1, use_graph=ug))
    rng = np.random.default_rng(1)
    for pn, shp in eng.param_shapes.items():
        eng.set_param(pn, rng.normal(0, 0.01, size=shp).astype(np.float32))
    bs = []
    for i in range(4):
        ids, vals, labels = synth_batch(B, 39, V, seed=100+i)
        si, sv, sl = eng.input_slot(i)
        si[:B].copy_(torch.from_numpy(ids)); sv[:B].copy_(torch.from_numpy(vals)); sl[:B].copy_(torch.from_numpy(labels))
        bs.append((si[:B], sv[:B], sl[:B]))
    for s in range(20): eng.train_step(*bs[s%4], want_loss=False)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for s in range(300): eng.train_step(*bs[s%4], want_loss=False)
    torch.cuda.synchronize(); el=time.perf_counter()-t0
    print("use_graph", ug, "%.4f ms/step" % (el/300*1e3), "%.0f ex/s" % (B*300/el))
    eng.close()
