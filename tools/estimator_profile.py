import cProfile, pstats, importlib.util, os, sys, tempfile, time
ROOT = "/root/repo" if os.path.exists("/root/repo/tools") else os.getcwd()
sys.path.insert(0, ROOT)
import numpy as np
from tf_repos_amd.synth import synth_batch
lines, epochs, F, V, B = 400000, 40, 39, 1_000_000, 4096
d = tempfile.mkdtemp(prefix="dctr_input_")
path = os.path.join(d, "tr.libsvm")
ids, vals, labels = synth_batch(lines, F, V, seed=5)
with open(path, "w") as f:
    for r in range(lines):
        f.write("%d " % labels[r] + " ".join("%d:%.6g" % (i, v) for i, v in zip(ids[r], vals[r])) + "\n")
import tf_repos_amd.tf_shim as shim
shim.install()
spec = importlib.util.spec_from_file_location("ctr_estimator_example", os.path.join(ROOT, "examples", "ctr_estimator.py"))
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
p = dict(model="deepfm", field_size=F, feature_size=V, embedding_size=16, learning_rate=5e-4, l2_reg=1e-4, deep_layers="400,400,400",
         dropout="0.5,0.5,0.5", cross_layers=3, optimizer="Adam")
est = mod.build_estimator(p, os.path.join(d, "ckpt"), log_steps=10 ** 9)
import torch
est.train(input_fn=lambda: mod.input_fn([path], num_epochs=1, batch_size=B))
torch.cuda.synchronize()
from tf_repos_amd import feeder as F_
acc = {}
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
    return w
F_.np.copyto = timed("feeder: np.copyto", np.copyto)
e = est._engine
e.input_slot_wait_released = timed("feeder: wait_released", e.input_slot_wait_released)
e.input_slot_fill = timed("feeder: fill", e.input_slot_fill)
_orig_run = F_.DeviceFeeder._run
def _run(self, it):
    def src():
        while True:
            t = time.perf_counter()
            try:
                x = next(it)
            except StopIteration:
                return
            acc["feeder: source next()"] = acc.get("feeder: source next()", 0.0) + time.perf_counter() - t
            yield x
    t = time.perf_counter()
    _orig_run(self, src())
    acc["feeder: thread total"] = time.perf_counter() - t
F_.DeviceFeeder._run = _run
pr = cProfile.Profile(); t0 = time.perf_counter()
if not os.environ.get("NO_CPROFILE"): pr.enable()
est.train(input_fn=lambda: mod.input_fn([path], num_epochs=epochs, batch_size=B))
torch.cuda.synchronize()
if not os.environ.get("NO_CPROFILE"): pr.disable()
dt = time.perf_counter() - t0
print("total %.2f s for %d steps" % (dt, lines * epochs // B))
for k, v in sorted(acc.items()): print("  %-28s %.3f s" % (k, v))
if not os.environ.get("NO_CPROFILE"): pstats.Stats(pr).sort_stats("cumulative").print_stats(8)
# the same engine driven directly (no feeder, no estimator loop)
e = est._engine
print("engine cfg:", {k: getattr(e.cfg, k) for k in ("model", "max_batch", "table_mode", "use_graph", "table_sweep_period", "dropout", "deep_layers", "optimizer", "l2_reg") if hasattr(e.cfg, k)})
bs = []
for i in range(4):
    a, b, c = synth_batch(B, F, V, seed=100 + i)
    si, sv, sl = e.input_slot(i)
    si[:B].copy_(torch.from_numpy(a)); sv[:B].copy_(torch.from_numpy(b)); sl[:B].copy_(torch.from_numpy(c))
    bs.append((si[:B], sv[:B], sl[:B]))
for s in range(20): e.train_step(*bs[s % 4], want_loss=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for s in range(400): e.train_step(*bs[s % 4], want_loss=False)
torch.cuda.synchronize(); print("direct: %.1f us/step" % (1e6 * (time.perf_counter() - t0) / 400))
