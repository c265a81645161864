"""How many host threads should the torch-CPU baseline use?  (run on the GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import deepctr_oracle as O
cfg = O.Config(model="deepfm", field_size=39, feature_size=1_000_000, embedding_size=16, deep_layers=(400, 400, 400), dropout=(0.5, 0.5, 0.5),
               l2_reg=1e-4, learning_rate=5e-4, optimizer="Adam")
print("cpu_count", os.cpu_count())
for nt in [8, 16, 32, 64, 128]:
    torch.set_num_threads(nt)
    p = O.init_params(cfg, seed=1, scale=0.01)
    opt = O.Optimizer(cfg, p)
    b = [O.synth_batch(4096, 39, 1_000_000, seed=i) for i in range(2)]
    O.train_step(cfg, p, opt, *b[0])
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 6 and n < 50:
        O.train_step(cfg, p, opt, *b[n % 2]); n += 1
    el = time.perf_counter() - t0
    print("threads %d: %.1f ms/step, %.0f examples/s" % (nt, 1e3 * el / n, 4096 * n / el), flush=True)
