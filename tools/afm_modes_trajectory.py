"""AFM at the reference's K = A = 256 (run.sh:18 shape, B = 1024): the same 60 Adagrad steps in gemm_mode split (tall split-precision products,
csrc/gemm_ts.h) and exact (f32 kernels) -- losses step by step and the variables at the end.  usage (GPU box): python tools/afm_modes_trajectory.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tf_repos_amd.engine import Engine, EngineConfig
from tf_repos_amd.synth import synth_batch

B, V, K, A, F, STEPS = 1024, 117581, 256, 256, 39, 60
out = {}
for mode in ("split", "exact"):
    eng = Engine(EngineConfig(model="afm", field_size=F, feature_size=V, embedding_size=K, deep_layers=(1,), dropout=(1.0, 1.0), attention_layers=(A,),
                              l2_reg=1e-4, learning_rate=1e-2, optimizer="Adagrad", max_batch=B, seed=1, use_graph=False, gemm_mode=mode))
    rng = np.random.default_rng(1)
    for pn, shp in eng.param_shapes.items():
        # (large enough that the attention network's gradients are not rounding noise: Adagrad's accumulator starts at 0.1)
        eng.set_param(pn, rng.normal(0, 0.15 if pn == "emb" else 0.1, size=shp).astype(np.float32))
    losses = []
    for s in range(STEPS):
        ids, vals, labels = synth_batch(B, F, V, seed=100 + s)
        t = [torch.from_numpy(a).cuda() for a in (ids, vals, labels)]
        losses.append(eng.train_step(*t))
    if mode == "split":
        init = {}
        rng2 = np.random.default_rng(1)
        for pn, shp in eng.param_shapes.items():
            init[pn] = rng2.normal(0, 0.15 if pn == "emb" else 0.1, size=shp).astype(np.float32)
    out[mode] = (np.array(losses), {k: eng.get_param(k) for k in eng.param_shapes})
    eng.close()
ls, le = out["split"][0], out["exact"][0]
print("step   loss split    loss exact    |diff|")
for s in list(range(0, STEPS, 10)) + [STEPS - 1]:
    print("%4d   %.7f    %.7f    %.2e" % (s, ls[s], le[s], abs(ls[s] - le[s])))
print("max |loss split - loss exact| over %d steps: %.2e" % (STEPS, float(np.abs(ls - le).max())))
for k in out["split"][1]:
    a, b = out["split"][1][k], out["exact"][1][k]
    d = np.abs(a - b)
    print("%-24s max |split - exact| %.2e, 99.9th percentile %.2e, mean %.2e  (moved from its initial value by up to %.2e, mean %.2e)" % (k, float(d.max()), float(np.quantile(d, 0.999)), float(d.mean()), float(np.abs(b - init[k].reshape(b.shape)).max()), float(np.abs(b - init[k].reshape(b.shape)).mean())))
