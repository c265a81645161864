"""Input pipeline at speed: libsvm text -> tensors -> GPU, c2 shape (39 fields, V=1e6, B=4096).
  1. dctr_parse_libsvm throughput (lines/s, MB/s) for 1 .. nproc threads on a synthetic Criteo-shaped file
  2. Estimator.train over that file (examples/ctr_estimator.py, DeepFM c2) -- parse once, then batches from memory through the
     DeviceFeeder: the end-to-end examples/s a user of the TF surface sees, next to bench.py's resident-input figure
usage (GPU box): python tools/input_bench.py [lines] [epochs]"""
import importlib.util
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from tf_repos_amd import input_pipeline as ip
from tf_repos_amd.synth import synth_batch

lines = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
F, V, B = 39, 1_000_000, 4096
d = tempfile.mkdtemp(prefix="dctr_input_")
path = os.path.join(d, "tr.libsvm")
ids, vals, labels = synth_batch(lines, F, V, seed=5)
t0 = time.perf_counter()
with open(path, "w") as f:
    for r in range(lines):
        f.write("%d " % labels[r] + " ".join("%d:%.6g" % (i, v) for i, v in zip(ids[r], vals[r])) + "\n")
size = os.path.getsize(path)
print(json.dumps({"file": "%d lines, %.1f MB (%.0f B/line), written in %.1f s" % (lines, size / 1e6, size / lines, time.perf_counter() - t0)}), flush=True)

for threads in sorted({1, 8, 32, os.cpu_count() or 8}):
    t0 = time.perf_counter()
    pi, pv, pl = ip.parse_file(path, F, threads=threads)
    dt = time.perf_counter() - t0
    assert len(pl) == lines and np.array_equal(pi, ids)
    print(json.dumps({"parse_threads": threads, "lines_per_sec": round(lines / dt), "MB_per_sec": round(size / dt / 1e6, 1)}), flush=True)

if os.environ.get("DCTR_INPUT_BENCH_PARSE_ONLY"):
    sys.exit(0)
import tf_repos_amd.tf_shim as shim
shim.install()
spec = importlib.util.spec_from_file_location("ctr_estimator_example", os.path.join(ROOT, "examples", "ctr_estimator.py"))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
p = dict(model="deepfm", field_size=F, feature_size=V, embedding_size=16, learning_rate=5e-4, l2_reg=1e-4, deep_layers="400,400,400",
         dropout="0.5,0.5,0.5", cross_layers=3, optimizer="Adam")
est = mod.build_estimator(p, os.path.join(d, "ckpt"), log_steps=10 ** 9)
import torch
est.train(input_fn=lambda: mod.input_fn([path], num_epochs=1, batch_size=B))            # parse + cache + engine creation
torch.cuda.synchronize()
t0 = time.perf_counter()
est.train(input_fn=lambda: mod.input_fn([path], num_epochs=epochs, batch_size=B))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
n = lines * epochs
print(json.dumps({"estimator_train": "DeepFM c2 from %s, %d epochs" % (os.path.basename(path), epochs), "examples_per_sec": round(n / dt),
                  "ms_per_step": round(1e3 * dt / (n / B), 4), "note": "includes graph trace + lowering, checkpoint save, H2D of every batch"}), flush=True)
