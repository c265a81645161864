"""Where does the fixed part of a timed bench block go?  Blocks of n steps (bench.py's loop: slots, hint, want_loss=False), each closed like the
bench's timed region (sync_tables + synchronize), for several n, twice; plus the flush alone by hipEvents.
usage (GPU box): python tools/fixed_cost_probe.py [split|exact]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf_repos_amd import capi                       # noqa: E402
from tf_repos_amd.engine import Engine, EngineConfig  # noqa: E402
from tf_repos_amd.synth import synth_batch          # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "split"
F, V, B, K = 39, 1_000_000, 4096, 16
eng = Engine(EngineConfig(model="deepfm", field_size=F, feature_size=V, embedding_size=K, deep_layers=(400, 400, 400), dropout=(0.5, 0.5, 0.5),
                          l2_reg=1e-4, learning_rate=5e-4, optimizer="Adam", max_batch=B, seed=1, gemm_mode=mode))
rng = np.random.default_rng(1)
for name, shp in eng.param_shapes.items():
    eng.set_param(name, rng.normal(0, 0.01, size=shp).astype(np.float32))
nb = capi.INPUT_SLOTS
slots = []
for i in range(nb):
    ids, vals, labels = synth_batch(B, F, V, seed=20260924 + 1 + i)
    si, sv, sl = eng.input_slot(i)
    si[:B].copy_(torch.from_numpy(ids)); sv[:B].copy_(torch.from_numpy(vals)); sl[:B].copy_(torch.from_numpy(labels))
    slots.append((si[:B], sv[:B], sl[:B]))
k = 0


def block(n, flush=True):
    global k
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.train_step(*slots[k % nb], want_loss=False)
        eng.prefetch_ids(slots[(k + 1) % nb][0])
        k += 1
    t_enq = time.perf_counter() - t0
    if flush:
        eng.sync_tables()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0), 1e3 * t_enq


block(5)
for rep in range(2):
    for n in (1, 2, 5, 10, 20, 50, 100, 200, 400):
        ms, enq = block(n)
        print("block of %3d steps + flush: %.3f ms total = %.4f ms/step (host enqueue %.3f ms)" % (n, ms, ms / n, enq))
for n in (20, 200):
    ms, enq = block(n, flush=False)
    print("block of %3d steps, NO flush (rows stay behind): %.3f ms total = %.4f ms/step" % (n, ms, ms / n))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.sync_tables(); e1.record(); e1.synchronize()
    print("   the flush alone (hipEvents): %.3f ms" % e0.elapsed_time(e1))
# idle gap before a block (the driver's run: engine build, then 5 + 20 steps)
for idle in (0.0, 0.05, 0.5):
    time.sleep(idle)
    ms, enq = block(20)
    print("after %.2f s idle: block of 20 + flush %.3f ms = %.4f ms/step" % (idle, ms, ms / 20))
eng.close()
