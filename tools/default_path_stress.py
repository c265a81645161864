"""Repeated-run check of the DEFAULT step path under load (round-4 verdict, weak #3): N times, the bench loop (lagging rows, next-batch hint,
input slots, dropout, deferred end-of-step join) against the classic sweep of the same engine, while a second engine on another thread keeps
the GPU busy with its own training steps (the condition under which the removed pre-advance showed its intermittent mismatch).
usage (GPU box): python tools/default_path_stress.py [runs=100] [gemm_mode=exact|split]"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf_repos_amd import capi                       # noqa: E402
from tf_repos_amd.engine import Engine, EngineConfig  # noqa: E402
from tf_repos_amd.synth import synth_batch          # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
mode = sys.argv[2] if len(sys.argv) > 2 else "exact"
F, V, B, K, STEPS = 39, 200_000, 2048, 16, 19
KW = dict(model="deepfm", field_size=F, feature_size=V, embedding_size=K, deep_layers=(400, 400), dropout=(0.5, 0.5), l2_reg=1e-4,
          learning_rate=5e-4, optimizer="Adam")
rng = np.random.default_rng(7)
P0 = None
BATCHES = [synth_batch(B, F, V, seed=500 + i) for i in range(capi.INPUT_SLOTS)]


def one(period, hint, seed):
    global P0
    eng = Engine(EngineConfig(max_batch=B, seed=seed, table_sweep_period=period, use_graph=False, gemm_mode=mode, **KW))
    if P0 is None:
        P0 = {n: rng.normal(0, 0.01, size=s).astype(np.float32) for n, s in eng.param_shapes.items()}
    eng.set_params(P0)
    slots = []
    for i, (ids, vals, labels) in enumerate(BATCHES):
        si, sv, sl = eng.input_slot(i)
        si[:B].copy_(torch.from_numpy(ids)); sv[:B].copy_(torch.from_numpy(vals)); sl[:B].copy_(torch.from_numpy(labels))
        slots.append((si[:B], sv[:B], sl[:B]))
    nb = len(slots)
    with torch.cuda.stream(eng.main_stream()):
        for s in range(STEPS):
            eng.train_step(*slots[s % nb], want_loss=(s == 11))
            if hint:
                eng.prefetch_ids(slots[(s + 1) % nb][0])
    out = dict(eng.get_params())
    out["emb/m"], out["emb/v"] = eng.get_slot("emb", 0), eng.get_slot("emb", 1)
    eng.check_ids()
    eng.close()
    return out


stop = threading.Event()


def load():
    e2 = Engine(EngineConfig(max_batch=4096, seed=9, use_graph=False, model="deepfm", field_size=F, feature_size=1_000_000, embedding_size=16,
                             deep_layers=(400, 400, 400), dropout=(0.5, 0.5, 0.5), l2_reg=1e-4, learning_rate=5e-4, optimizer="Adam"))
    ids, vals, labels = synth_batch(4096, F, 1_000_000, seed=3)
    t = [torch.from_numpy(a).cuda() for a in (ids, vals, labels)]
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        while not stop.is_set():
            for _ in range(20):
                e2.train_step(*t, want_loss=False)
            st.synchronize()
    e2.close()


th = threading.Thread(target=load, daemon=True)
th.start()
ref = one(1, False, 1)
fails, worst = 0, 0.0
t0 = time.time()
for r in range(runs):
    got = one(0, True, 1)
    err = max(float(np.abs(got[k] - v).max()) / (max(float(np.abs(v).max()), 1e-30) / 5e-4 if k.endswith(("/m", "/v")) else 1.0) for k, v in ref.items())
    n_off = sum(int((np.abs(got[k] - v) > 1e-6).sum()) for k, v in ref.items() if not k.endswith(("/m", "/v")))
    worst = max(worst, err)
    # (a one-unit ReLU flip -- tests/test_bench_path_gpu.py -- shows as <= 5e-5 in a few hundred elements; a race in the table path as ~lr = 5e-4)
    if err > 5e-5 or n_off > 2000:
        fails += 1
        print("run %d: MISMATCH max %.2e, %d elements > 1e-6" % (r, err, n_off), flush=True)
stop.set()
th.join(timeout=30)
print("%d of %d runs of the default path (gemm_mode %s, %d steps, lag + hint + slots + dropout + deferred join, a second engine training beside it) "
      "differ from the classic sweep by more than 5e-5; worst max difference %.2e; %.0f s" % (fails, runs, mode, STEPS, worst, time.time() - t0))
sys.exit(1 if fails else 0)
