"""The C ABI from plain C (no Python, no torch in the host program): examples/c_abi_train.c is compiled with gcc against
include/deepctr_hip.h + libdeepctr_hip.so and trains a DeepFM for three steps; the losses it prints must be the oracle's on the
same inputs (regenerated here from the program's xorshift stream)."""
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import deepctr_oracle as O
from tf_repos_amd import capi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class XorShift:
    def __init__(self):
        self.s = 88172645463325252

    def u32(self):
        m = (1 << 64) - 1
        self.s ^= (self.s << 13) & m
        self.s ^= self.s >> 7
        self.s ^= (self.s << 17) & m
        return self.s >> 32

    def unit(self):
        return np.float32((self.u32() >> 8) * (1.0 / 16777216.0))


def test_c_program_trains_like_the_oracle(tmp_path, dev):
    exe = str(tmp_path / "c_abi_train")
    libdir = os.path.dirname(capi.LIB_PATH)
    cmd = ["gcc", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_train.c"), "-o", exe, "-L" + libdir,
           "-ldeepctr_hip", "-Wl,-rpath," + libdir, "-lm"]
    subprocess.run(cmd, check=True)
    out = subprocess.run([exe, "3"], check=True, capture_output=True, text=True, timeout=120).stdout
    got = [float(l.split()[-1]) for l in out.splitlines() if l.startswith("step")]
    assert len(got) == 3

    B, F, K, V = 64, 39, 8, 2000
    ocfg = O.Config(model="deepfm", field_size=F, feature_size=V, embedding_size=K, deep_layers=(32, 16), dropout=(1.0, 1.0), l2_reg=1e-3,
                    learning_rate=1e-2, optimizer="Adagrad")
    rng = XorShift()
    # parameters in the engine's declaration order (what dctr_param_info enumerates)
    from tf_repos_amd.engine import Engine, EngineConfig
    eng = Engine(EngineConfig(model="deepfm", field_size=F, feature_size=V, embedding_size=K, deep_layers=(32, 16), dropout=(1.0, 1.0),
                              l2_reg=1e-3, learning_rate=1e-2, optimizer="Adagrad", max_batch=B))
    params = {}
    for name, shp in eng.param_shapes.items():
        n = int(np.prod(shp))
        params[name] = torch.from_numpy(np.array([np.float32(0.1) * (rng.unit() - np.float32(0.5)) for _ in range(n)], dtype=np.float32).reshape(shp))
    eng.close()
    ids = np.empty((B, F), np.int32)
    vals = np.empty((B, F), np.float32)
    for i in range(B * F):
        ids.flat[i] = rng.u32() % V
        vals.flat[i] = rng.unit()
    labels = np.array([1.0 if rng.unit() < np.float32(0.3) else 0.0 for _ in range(B)], np.float32)
    opt = O.Optimizer(ocfg, params)
    for s in range(3):
        ref, _ = O.train_step(ocfg, params, opt, ids, vals, labels)
        assert abs(got[s] - ref) <= 1e-5 * max(1.0, abs(ref)), (s, got[s], ref)
