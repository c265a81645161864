"""Adam's update term lr_t m / (sqrt(v) + eps) in the TABLE kernels (dense-exact sweep, replay of lagging rows, fused tail) uses
v_sqrt_f32 / v_rcp_f32 (1 ulp each) by default; DCTR_IEEE_ADAM=1 in the environment loads the second library that
__graft_entry__.build() makes from the same sources with -DDCTR_IEEE_ADAM: the correctly rounded sqrtf and division there too
(csrc/opt_rules.h).  The dense arena (MLP / cross / attention weights) always takes the correctly rounded forms.
Here: the parity suites that pin Adam pass on BOTH libraries (child processes: the library is chosen at import), and the two modes
agree with each other to 1e-6 after 40 steps (lr 1e-2: forty updates of ~1e-2 each, a few ulp apart per step) of lagging rows while not being bit-identical (the knob does something)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np, torch
from oracle import deepctr_oracle as O
from tests.util import dev_batch
from tf_repos_amd.engine import Engine, EngineConfig
dev = torch.device("cuda:0")
F, V, B, K = 39, 20000, 256, 16
kw = dict(model="deepfm", field_size=F, feature_size=V, embedding_size=K, deep_layers=(64, 32), dropout=(1.0, 1.0), l2_reg=1e-3, learning_rate=1e-2, optimizer="Adam")
ocfg = O.Config(**kw)
params = O.init_params(ocfg, seed=3, scale=0.05)
eng = Engine(EngineConfig(max_batch=B, seed=1, table_sweep_period=8, **kw))
eng.set_params(params)
for s in range(40):
    eng.train_step(*dev_batch(*O.synth_batch(B, F, V, seed=4000 + s), dev), want_loss=False)
np.savez(sys.argv[1], emb=eng.get_param("emb"), lin=eng.get_param("linear"), w=eng.get_param("mlp0/weights"), m=eng.get_slot("emb", 0))
"""


def _child(args, ieee, timeout=900):
    env = dict(os.environ, DCTR_IEEE_ADAM="1" if ieee else "0")
    return subprocess.run([sys.executable] + args, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("ieee", [True])
def test_adam_parity_suites_with_ieee_update_term(ieee, dev):
    """(the default mode runs in the suites themselves)"""
    r = _child(["-m", "pytest", "-q", "-m", "gpu", "-x", "tests/test_lag_gpu.py::test_lagging_rows_match_the_oracle",
                "tests/test_lag_gpu.py::test_restored_global_step_and_written_parameters", "tests/test_model_golden.py", "-k",
                "lagging or restored or adam or deepfm_c1 or dcn or dropout"], ieee)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-1000:]


def test_the_two_modes_agree_but_are_not_the_same_code(dev, tmp_path):
    out = {}
    for ieee in (False, True):
        f = str(tmp_path / ("ieee%d.npz" % ieee))
        r = _child(["-c", CHILD, f], ieee)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        out[ieee] = dict(np.load(f))
    same = True
    for k in out[False]:
        d = float(np.abs(out[False][k] - out[True][k]).max())
        same = same and d == 0.0
        assert d <= 1e-6, (k, d)
    assert not same, "DCTR_IEEE_ADAM=1 changed nothing: is the knob wired?"
