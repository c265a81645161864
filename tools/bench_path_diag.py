#!/usr/bin/env python
"""Diagnostic behind tests/test_bench_path_gpu.py: c2 for N steps under several engine variants, each against the fp64 oracle, with
where the largest differences sit.  python tools/bench_path_diag.py [steps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import deepctr_oracle as O
from tf_repos_amd import capi
from tf_repos_amd.engine import Engine, EngineConfig
from tf_repos_amd.synth import synth_batch

F, V, B, K = 39, 1_000_000, 4096, 16
layers = (400, 400, 400)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 17
keep_all = float(os.environ.get("DIAG_KEEP", "0.5"))
keep = tuple(keep_all for _ in layers)
kw = dict(model="deepfm", field_size=F, feature_size=V, embedding_size=K, deep_layers=layers, dropout=keep, l2_reg=1e-4, learning_rate=5e-4,
          optimizer="Adam")
ocfg = O.Config(**kw)
p0 = O.init_params(ocfg, seed=20260925, scale=0.01)
host = [synth_batch(B, F, V, seed=20260924 + 1 + i) for i in range(8)]


def masks_for(eng, step):
    return {"mlp%d" % i: torch.from_numpy(eng.dropout_mask(capi.SITE_MLP(i), (B, h), keep[i], step=step).astype(np.float32)) for i, h in enumerate(layers)}


def run_engine(period, hint):
    eng = Engine(EngineConfig(max_batch=B, seed=1, table_sweep_period=period, use_graph=False, **kw))
    eng.set_params(p0)
    slots = []
    for i in range(8):
        si, sv, sl = eng.input_slot(i)
        si[:B].copy_(torch.from_numpy(host[i][0])); sv[:B].copy_(torch.from_numpy(host[i][1])); sl[:B].copy_(torch.from_numpy(host[i][2]))
        slots.append((si[:B], sv[:B], sl[:B]))
    for s in range(steps):
        eng.train_step(*slots[s % 8], want_loss=False)
        if hint:
            eng.prefetch_ids(slots[(s + 1) % 8][0])
    got = dict(eng.get_params())
    got["emb/m"], got["emb/v"] = eng.get_slot("emb", 0), eng.get_slot("emb", 1)
    eng.close()
    return got


eng0 = Engine(EngineConfig(max_batch=B, seed=1, **kw))
all_masks = [masks_for(eng0, s + 1) for s in range(steps)]
eng0.close()
truth = {}
for dt in (torch.float64, torch.float32):
    p = {k: v.to(dt) for k, v in p0.items()}
    opt = O.Optimizer(ocfg, p)
    for s in range(steps):
        mk = {k: v.to(dt) for k, v in all_masks[s].items()}
        if dt == torch.float64:        # pre-activations within rounding of zero: whose ReLU mask the last bit decides
            ids, vals, _ = host[s % 8]
            x = (p["emb"][torch.from_numpy(ids).long()] * torch.from_numpy(vals).to(dt)[:, :, None]).reshape(B, F * K)
            for i in range(len(layers)):
                z = x @ p["mlp%d/weights" % i] + p["mlp%d/biases" % i]
                mn, idx = z.abs().reshape(-1).topk(2, largest=False)
                for v, j in zip(mn, idx):
                    if float(v) < 3e-9:
                        print("   step %d layer %d: |z| = %.1e at example %d, unit %d" % (s, i, float(v), int(j) // layers[i], int(j) % layers[i]))
                x = torch.relu(z) * mk["mlp%d" % i] / keep[i]
        O.train_step(ocfg, p, opt, *host[s % 8], masks=mk)
    truth[dt] = {k: v.numpy().astype(np.float64) for k, v in p.items()}
    truth[dt]["emb/m"] = opt.slots["emb"]["m"].numpy().astype(np.float64)
    truth[dt]["emb/v"] = opt.slots["emb"]["v"].numpy().astype(np.float64)
t64 = truth[torch.float64]
print("steps %d keep %s" % (steps, keep))
print("fp32 oracle vs fp64:", {k: "%.2e" % np.abs(truth[torch.float32][k] - v).max() for k, v in t64.items()})
hot = set(range(1, 14))
for name, period, hint in (("lag8+hint", 0, True), ("classic+hint", 1, True), ("classic", 1, False), ("lag8", 0, False)):
    got = run_engine(period, hint)
    print(name, {k: "%.2e" % np.abs(got[k].astype(np.float64) - v).max() for k, v in t64.items()})
    for k in ("emb", "mlp0/weights", "mlp1/weights"):
        d = np.abs(got[k].astype(np.float64) - t64[k])
        idx = np.argsort(d.reshape(-1))[::-1][:8]
        rows, cols = np.unravel_index(idx, d.shape)
        print("   %s worst at" % k, [(int(r), int(c), "%.1e" % d[r, c]) for r, c in zip(rows, cols)], "n>5e-6:", int((d > 5e-6).sum()))
