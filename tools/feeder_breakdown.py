"""Where a DeviceFeeder-driven step spends its host time (c2 shape): the consumer's wait for a staged batch, the train_step enqueue, the
slot release and the next-batch hint; on the feeder thread the copy into pinned memory and the three H2D enqueues.
usage (GPU box): python tools/feeder_breakdown.py [steps]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tf_repos_amd.engine import Engine, EngineConfig
from tf_repos_amd import feeder as F_
from tf_repos_amd.synth import synth_batch
B, F, V = 4096, 39, 1_000_000
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
eng = Engine(EngineConfig(model="deepfm", field_size=F, feature_size=V, embedding_size=16, deep_layers=(400, 400, 400), dropout=(0.5, 0.5, 0.5),
                          l2_reg=1e-4, learning_rate=5e-4, optimizer="Adam", max_batch=B, seed=1))
NSRC = int(os.environ.get("PROBE_DISTINCT", "6"))
src = [synth_batch(B, F, V, seed=10 + i) for i in range(NSRC)]
def gen():
    for s in range(steps):
        yield src[s % NSRC]
    if os.environ.get("PROBE_PARTIAL"):
        yield tuple(a[:2688] for a in src[0])

acc = {}
def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return w
_copyto = np.copyto
F_.np.copyto = timed("feeder: np.copyto into pinned", _copyto)
eng.input_slot_wait_released = timed("feeder: wait_released (slot's step done on the device)", eng.input_slot_wait_released)
eng.input_slot_fill = timed("feeder: fill (3 H2D enqueues + record)", eng.input_slot_fill)

if os.environ.get("PROBE_NO_EVENTS"):          # timing experiment only (races): no device-side wait, no consumed record
    eng.input_slot_acquire = lambda k, stream=None: None
    eng.input_slot_release = lambda k, stream=None: None
if os.environ.get("PROBE_SIDE_STREAM"):        # the steps on a torch side stream instead of the default stream
    torch.cuda.set_stream(torch.cuda.Stream())
for rep in range(2):
    acc.clear()
    fd = F_.DeviceFeeder(eng, gen())
    it = iter(fd)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    while True:
        a = time.perf_counter()
        try:
            ids, vals, labels, k = next(it)
        except StopIteration:
            break
        b = time.perf_counter()
        eng.train_step(ids, vals, labels, want_loss=False)
        c = time.perf_counter()
        fd.release(k)
        d = time.perf_counter()
        nxt = fd.peek_next_ids(wait=0.05)
        if nxt is not None:
            eng.prefetch_ids(nxt)
        e = time.perf_counter()
        for nm, v in (("consumer: next batch (queue wait + wait_event + views)", b - a), ("consumer: train_step enqueue", c - b),
                      ("consumer: release", d - c), ("consumer: peek + prefetch_ids", e - d)):
            acc[nm] = acc.get(nm, 0.0) + v
        n += 1
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    fd.close()
    print("rep %d: %d steps, %.1f us/step wall (%.2f M examples/s)" % (rep, n, 1e6 * dt / n, n * B / dt / 1e6))
    for nm, v in sorted(acc.items()):
        print("   %-60s %7.1f us/step" % (nm, 1e6 * v / n))
