"""Step time of the CSR (multi-hot) models at the Ali-CCP shape of DeepMTL/README.md:19-25,38 (V ~ 4.5 M ids, ~250 ids per
example): DIN (sum pooling), ESMM and DIN with attention pooling (att_fc0 = 256 wide, DIN.py:164), MLP 256-128-64 (DIN.py:40 / DeepCvrMTL.py:51), K=16.  One JSON line per configuration.
usage (GPU box): python tools/multihot_bench.py [steps] [B]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from tf_repos_amd.engine import Engine, EngineConfig

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
V, K, FC = 4_500_000, 16, 11
S = FC + 8
dev = torch.device("cuda", 0)
rng = np.random.default_rng(20260924)


def synth(B, seed):
    """slot-ordered CSR: 11 one-hot fields, 4 weighted multi-hot user features (~60 ids each, Zipf over a 1M-id range each),
    3 single ad ids, 1 unweighted multi-hot ad feature (~5 ids)"""
    r = np.random.default_rng(seed)
    lens = np.ones((B, S), np.int64)
    lens[:, FC:FC + 4] = r.poisson(60, size=(B, 4))
    lens[:, S - 1] = r.poisson(5, size=B)
    off = np.concatenate([[0], np.cumsum(lens.ravel())]).astype(np.int32)
    nnz = int(off[-1])
    seg_slot = np.repeat(np.tile(np.arange(S), B), lens.ravel())
    base = (seg_slot.astype(np.int64) * (V // S))
    ids = (base + np.minimum(r.zipf(1.1, size=nnz), V // S - 1)).astype(np.int32)
    wts = np.where((seg_slot >= FC) & (seg_slot < FC + 4), r.uniform(0.5, 3.0, size=nnz), 1.0).astype(np.float32)
    y = (r.random(B) < 0.04).astype(np.float32)
    z = (y * (r.random(B) < 0.05)).astype(np.float32)
    return off, ids, wts, y, z


batches = [[torch.from_numpy(a).to(dev) for a in synth(B, 100 + i)] for i in range(4)]
max_nnz = max(int(b[1].shape[0]) for b in batches)
for model in ("din", "esmm", "din_att"):
    for table_mode in ("dense_exact", "touched_rows"):
        if os.environ.get("DCTR_MH_ONLY") and os.environ["DCTR_MH_ONLY"] != "%s:%s" % (model, table_mode):
            continue
        att = dict(attention_layers=(256,), att_pairs=[(FC + i, FC + 4 + i) for i in range(4)]) if model == "din_att" else {}
        name, model = model, model.split("_")[0]
        eng = Engine(EngineConfig(model=model, field_size=S, feature_size=V, embedding_size=K, deep_layers=(256, 128, 64), dropout=(0.5, 0.5, 0.5),
                                  l2_reg=1e-4, learning_rate=5e-4, optimizer="Adam", table_mode=table_mode, max_batch=B,
                                  max_entries=max_nnz + 1024, seed=1, **att))
        for pn, shp in eng.param_shapes.items():
            if pn != "emb":
                eng.set_param(pn, rng.normal(0, 0.01, size=shp).astype(np.float32))
        for i in range(5):
            off, ids, wts, y, z = batches[i % 4]
            eng.train_step_csr(off, ids, wts, y, z if model == "esmm" else None, want_loss=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            off, ids, wts, y, z = batches[i % 4]
            eng.train_step_csr(off, ids, wts, y, z if model == "esmm" else None, want_loss=False)
        eng.sync_tables()          # (time-blocked table sweep: the timed steps' updates are all computed inside the timed region)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        eng.check_ids()
        model = name
        print(json.dumps({"config": "%s B=%d V=4.5e6 K=16 slots=%d MLP 256-128-64 avg nnz/example=%.0f table=%s" % (model, B, S, max_nnz / B, table_mode),
                          "ms_per_step": round(dt * 1e3, 4), "examples_per_sec": round(B / dt, 1), "ids_per_sec": round(max_nnz / dt, 1)}), flush=True)
        eng.close()
