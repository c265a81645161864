"""The judged gather figure (bench.py kernels.embed_gather_fwd_k32_hbm: K = 32, 32 M-row table, uniform ids, B = 4096, the op through the C ABI,
back-to-back launches) under the kernel's A/B knobs -- one child process per variant (the knobs are read once).
usage (GPU box): python tools/gather_k32_probe.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys
sys.path.insert(0, %r)
import torch, bench
from tf_repos_amd.engine import Engine
dev = torch.device("cuda", 0)
copy = Engine.measure_copy_bandwidth(1 << 30, 20)
for K, V in ((32, 32 * 1024 * 1024), (16, 64 * 1024 * 1024)):
    best = None
    for rep in range(3):
        ms, nbytes = bench.hbm_resident_gather(dev, K=K, V=V, B=4096, F=39)
        best = ms if best is None or ms < best else best
    print("K=%%d: %%.2f us  %%.0f GB/s algorithmic = %%.3f of the measured copy rate (%%.0f GB/s)" %% (K, best * 1e3, nbytes / best / 1e6, nbytes / best / 1e6 / copy, copy))
""" % ROOT
for k32 in ("1", "0", "2"):
    for nt in ("0", "1"):
        env = dict(os.environ, DCTR_GATHER_K32=k32, DCTR_GATHER_NT=nt)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
        for line in (r.stdout.strip().splitlines() or [r.stderr[-300:]]):
            print("lane mapping %s (0: <8,4,5>, 1: <8,8,5>, 2: <8,4,10>)  nt stores %s | %s" % (k32, nt, line), flush=True)
