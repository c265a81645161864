"""AFM at the reference's own operating point (run.sh:18: `AFM.py --optimizer=Adam --learning_rate=0.0005 --batch_size=128
--embedding_size=256 --attention_layers=128 --dropout=0.5,0.5 --l2_reg=0.001`), compared at the level of GRADIENTS.

Why gradients: Adam's update lr_t m / (sqrt(v) + eps) turns the fp32 rounding of a ~1e-8 gradient into a visible fraction of lr, and
AFM's attention network sits behind a softmax over 741 pairs: most of its gradient elements ARE that small.  The other K = 256 tests
therefore train with Adagrad; this one keeps the reference's optimizer and checks what the engine computes BEFORE Adam amplifies it:
after the first Adam step from zero slots m = (1 - beta1) g exactly (DeepFM.py:205 / AFM.py:185 rule), so `dctr_slot_get(name, 0)`
IS the gradient the step saw -- for every dense variable and for every row of the embedding / linear tables (the dense-exact table
gradient: UnsortedSegmentSum + l2 theta, AFM.py:180-181).

Yardstick: the fp64 evaluation of the oracle's autograd on the same inputs and dropout masks is the truth; the oracle evaluated in fp32
shows what fp32 arithmetic costs on this graph.  The engine's fp32 gradient must be as close to the truth as that (4x slack for a
different summation order), or within 1e-6 of the variable's largest gradient element.  Then: four more Adam steps, every variable
compared on the elements whose gradient is not rounding noise (|g| >= 1e-4 of ... the golden tests' rule, tests/test_model_golden.py
var_err)."""
import numpy as np
import pytest
import torch

from oracle import deepctr_oracle as O
from tests.test_dropout_gpu import oracle_masks
from tests.util import dev_batch
from tf_repos_amd.engine import Engine, EngineConfig

pytestmark = pytest.mark.gpu


def test_afm_gradients_at_the_run_sh_operating_point(dev):
    F, V, B, K, A = 39, 20000, 128, 256, 128
    kw = dict(model="afm", field_size=F, feature_size=V, embedding_size=K, deep_layers=(1,), attention_layers=(A,), dropout=(0.5, 0.5),
              l2_reg=1e-3, learning_rate=5e-4, optimizer="Adam")
    ocfg = O.Config(**kw)
    p32 = O.init_params(ocfg, seed=18, scale=0.02)
    eng = Engine(EngineConfig(max_batch=B, seed=77, use_graph=False, **kw))
    eng.set_params(p32)
    ids, vals, labels = O.synth_batch(B, F, V, seed=1800)
    masks = oracle_masks(eng, ocfg, B)
    m64 = {k: v.double() for k, v in masks.items()}
    p64 = {k: v.double() for k, v in p32.items()}
    _, g64, _ = O.grads(ocfg, p64, ids, vals, labels, train=True, masks=m64)
    _, g32, _ = O.grads(ocfg, p32, ids, vals, labels, train=True, masks=masks)
    oopt = O.Optimizer(ocfg, p32)
    ref_loss, _ = O.train_step(ocfg, p32, oopt, ids, vals, labels, masks=masks)
    loss = eng.train_step(*dev_batch(ids, vals, labels, dev))
    assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss))
    one_minus_b1 = float(np.float32(1.0) - np.float32(0.9))
    report = {}
    for name, gt in g64.items():
        truth = gt.numpy()
        g_eng = eng.get_slot(name, 0).astype(np.float64) / one_minus_b1
        scale = float(np.abs(truth).max())
        err_eng = float(np.abs(g_eng - truth).max())
        err_o32 = float(np.abs(g32[name].double().numpy() - truth).max())
        report[name] = (err_eng / scale, err_o32 / scale)
        if scale < 1e-18:
            # attention_out/biases: the softmax over the pairs is shift-invariant, so this gradient is identically zero (fp64: 4e-23) and what
            # an fp32 evaluation returns is the rounding residue of sum(d score) over the 94 848 pair rows (|d score| ~ 1e-9: eps * 1e-9 *
            # sqrt(rows) ~ 2e-14) -- a property of the summation order, not of the model: bounded, not compared with the oracle's own residue
            assert err_eng <= 1e-12, (name, err_eng, err_o32, scale)
            continue
        assert err_eng <= max(4.0 * err_o32, 1e-6 * scale), (name, err_eng, err_o32, scale)
        if name not in ("emb", "linear"):       # the same gradient read directly (dctr_param_grad_get: the partial slabs of the backward, summed)
            g_direct = eng.get_grad(name).astype(np.float64).reshape(truth.shape)
            l2_term = 0.0                       # (AFM's loss regularises the two tables only, AFM.py:180-181)
            assert float(np.abs(g_direct + l2_term - truth).max()) <= max(4.0 * err_o32, 1e-6 * scale), name
    print("AFM K=256 A=128 Adam, |g_engine - g_fp64| / max|g| (fp32 oracle beside it):",
          {k: "%.1e (%.1e)" % v for k, v in report.items()})
    # four more steps under Adam; elements whose gradient was rounding noise at any step are left out (Adam amplifies their noise in
    # TF's fp32 as much as here), everything else must agree like the other models do
    live = {name: (g.abs() >= 1e-4 * g.abs().max()).numpy() for name, g in g64.items()}
    for step in range(1, 5):
        ids, vals, labels = O.synth_batch(B, F, V, seed=1800 + step)
        masks = oracle_masks(eng, ocfg, B)
        _, g, _ = O.grads(ocfg, p32, ids, vals, labels, train=True, masks=masks)
        for name in live:
            live[name] &= (g[name].abs() >= 1e-4 * g[name].abs().max()).numpy()
        ref_loss, _ = O.train_step(ocfg, p32, oopt, ids, vals, labels, masks=masks)
        loss = eng.train_step(*dev_batch(ids, vals, labels, dev))
        assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (step, loss, ref_loss)
    got = eng.get_params()
    for name, ref in p32.items():
        d = np.abs(got[name] - ref.numpy())
        keep = live.get(name)
        frac = float(keep.mean()) if keep is not None else 1.0
        worst = float(d[keep].max()) if keep is not None and keep.any() else (float(d.max()) if keep is None else 0.0)
        print("  %-22s compared %.1f %% of the elements, max |diff| %.2e (all elements: %.2e)" % (name, 100 * frac, worst, float(d.max())))
        assert worst <= 5e-6, (name, worst)
    eng.close()
