"""Step time of the other BASELINE configs (parity-test cases, not the headline bench): c1 DeepFM B=256 V=117581 K=8, c3 DCN,
c4 PNN(inner) / NFM at B=8192 K=32, plus AFM and DeepMVM at c2's shape.  One JSON line per config.
usage (GPU box): python tools/config_bench.py [steps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from tf_repos_amd.engine import Engine, EngineConfig
from tf_repos_amd.synth import synth_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
only = sys.argv[2] if len(sys.argv) > 2 else None
CONFIGS = {
    "c1 DeepFM B=256 V=117581 K=8 MLP 400x3": dict(model="deepfm", B=256, V=117581, K=8, layers=(400, 400, 400)),
    "c2 DeepFM B=4096 V=1e6 K=16 MLP 400x3": dict(model="deepfm", B=4096, V=1_000_000, K=16, layers=(400, 400, 400)),
    "c3 DCN B=4096 V=1e6 K=16 cross 3 MLP 400x2": dict(model="dcn", B=4096, V=1_000_000, K=16, layers=(400, 400), cross=3),
    "c4 PNN-inner B=8192 V=1e6 K=32 MLP 256-128": dict(model="ipnn", B=8192, V=1_000_000, K=32, layers=(256, 128)),
    "c4 NFM B=8192 V=1e6 K=32 MLP 256-128": dict(model="nfm", B=8192, V=1_000_000, K=32, layers=(256, 128)),
    "c4 PNN-outer B=8192 V=1e6 K=32 MLP 256-128 (P*K^2 = 758784 pair products per example, formed inside the first-layer GEMMs)":
        dict(model="opnn", B=8192, V=1_000_000, K=32, layers=(256, 128), steps=5),
    "AFM B=4096 V=1e6 K=16 att 256": dict(model="afm", B=4096, V=1_000_000, K=16, layers=(1,), att=(256,)),
    "DeepMVM B=4096 V=1e6 K=16 MLP 400x3": dict(model="mvm", B=4096, V=1_000_000, K=16, layers=(400, 400, 400)),
    # the reference's own AFM operating point (AFM.py:44,52 / run.sh:18): K = 256, attention 256 -- the unfused attention path
    "AFM reference point B=128 V=117581 K=256 att 256": dict(model="afm", B=128, V=117581, K=256, layers=(1,), att=(256,)),
    "AFM run.sh:18 point B=128 V=117581 K=256 att 128": dict(model="afm", B=128, V=117581, K=256, layers=(1,), att=(128,)),
    "AFM run.sh:18 point B=256 V=117581 K=256 att 128": dict(model="afm", B=256, V=117581, K=256, layers=(1,), att=(128,)),
    "AFM run.sh:18 point B=512 V=117581 K=256 att 128": dict(model="afm", B=512, V=117581, K=256, layers=(1,), att=(128,)),
    "AFM run.sh:18 point B=4096 V=117581 K=256 att 128": dict(model="afm", B=4096, V=117581, K=256, layers=(1,), att=(128,), steps=30),
    "AFM reference point B=512 V=117581 K=256 att 256": dict(model="afm", B=512, V=117581, K=256, layers=(1,), att=(256,)),
    "AFM reference point B=1024 V=117581 K=256 att 256": dict(model="afm", B=1024, V=117581, K=256, layers=(1,), att=(256,), steps=30),
    "AFM reference point B=4096 V=117581 K=256 att 256": dict(model="afm", B=4096, V=117581, K=256, layers=(1,), att=(256,), steps=10),
}
dev = torch.device("cuda", 0)
for name, c in CONFIGS.items():
    if only is not None and only not in name:
        continue
    keep = (0.5,) * max(len(c["layers"]), 2)
    eng = Engine(EngineConfig(model=c["model"], field_size=39, feature_size=c["V"], embedding_size=c["K"], deep_layers=c["layers"],
                              dropout=keep, cross_layers=c.get("cross", 3), attention_layers=c.get("att", (256,)), l2_reg=1e-4,
                              learning_rate=5e-4, optimizer="Adam", max_batch=c["B"], seed=1,
                              use_graph=os.environ.get("DCTR_USE_GRAPH", "0") == "1"))
    rng = np.random.default_rng(1)
    for pn, shp in eng.param_shapes.items():
        eng.set_param(pn, rng.normal(0, 0.01, size=shp).astype(np.float32))
    batches = []
    for i in range(4):
        ids, vals, labels = synth_batch(c["B"], 39, c["V"], seed=100 + i)
        si, sv, sl = eng.input_slot(i)
        si[:c["B"]].copy_(torch.from_numpy(ids)); sv[:c["B"]].copy_(torch.from_numpy(vals)); sl[:c["B"]].copy_(torch.from_numpy(labels))
        batches.append((si[:c["B"]], sv[:c["B"]], sl[:c["B"]]))
    steps_c = c.get("steps", steps)
    hint = os.environ.get("DCTR_CFG_HINT", "1") == "1"        # announce the next batch's ids after every step, as the input pipeline does
    # on the engine's own stream, as tf_shim's Estimator issues its steps (at c1 the legacy default stream costs 0.17 vs 0.13 ms/step;
    # at the large-batch configs it makes no difference); DCTR_CFG_MAIN_STREAM=0: torch's current stream
    import contextlib
    torch.cuda.synchronize()
    with (torch.cuda.stream(eng.main_stream()) if os.environ.get("DCTR_CFG_MAIN_STREAM", "1") == "1" else contextlib.nullcontext()):
        for s in range(min(10, steps_c)):
            eng.train_step(*batches[s % 4], want_loss=False)
            if hint: eng.prefetch_ids(batches[(s + 1) % 4][0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(steps_c):
            eng.train_step(*batches[s % 4], want_loss=False)
            if hint: eng.prefetch_ids(batches[(s + 1) % 4][0])
        eng.sync_tables()          # (time-blocked table sweep: every row's updates of the timed steps computed inside the timed region)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    print(json.dumps({"config": name, "ms_per_step": round(1e3 * el / steps_c, 4), "examples_per_sec": round(c["B"] * steps_c / el, 1)}), flush=True)
    eng.close()
