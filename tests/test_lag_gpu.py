"""Time-blocked dense-exact table sweep (csrc/lag.h, dctr_config.table_sweep_period): rows no batch touches may lag behind
global_step and are advanced through the steps they missed -- the same Adam update calls with the same per-step lr_t, in order --
when something reads them.  The scheme changes WHEN a row's updates are computed, never what they are, so a lagging engine must
end where the classic one (every row every step, period 1) ends: compared here element for element at 3e-6 or 4x what two runs of the
SAME engine differ by (the hot ids' segment sums meet through float atomics in no fixed order); a missed, doubled or mis-stamped
step would show as ~lr = 1e-2 -- and against the oracle's dense Adam (DeepFM.py:188-190,204-213) at the usual 2e-6."""
import numpy as np
import pytest
import torch

from oracle import deepctr_oracle as O
from tests.util import dev_batch
from tf_repos_amd import errors
from tf_repos_amd.engine import Engine, EngineConfig

pytestmark = pytest.mark.gpu


def engine(model, period, V, B, K=8, layers=(32, 16), F=39, seed=5, l2=1e-3, lr=1e-2, keep=None, params=None, cross=2):
    keep = keep or tuple(1.0 for _ in layers)
    eng = Engine(EngineConfig(model=model, field_size=F, feature_size=V, embedding_size=K, deep_layers=layers, dropout=keep, cross_layers=cross,
                              l2_reg=l2, learning_rate=lr, optimizer="Adam", max_batch=B, seed=seed, table_sweep_period=period))
    if params is not None:
        eng.set_params(params)
    return eng


def state_of(eng):
    out = dict(eng.get_params())
    for name in ("emb", "linear"):
        if name in eng.param_shapes:
            out[name + "/m"], out[name + "/v"] = eng.get_slot(name, 0), eng.get_slot(name, 1)
    return out


@pytest.mark.parametrize("model", ["deepfm", "dcn", "nfm"])
@pytest.mark.parametrize("period", [2, 5, 8])
def test_lagging_rows_end_where_the_classic_sweep_ends(model, period, dev):
    """30 steps, most of them without a loss read (rows lag up to `period` steps), with a loss-reporting step, a predict and a
    parameter read in the middle (each brings the rows to the present), a short last batch, dropout on: classic == lagging."""
    F, V, B, K = 39, 20000, 192, 8
    ocfg = O.Config(model=model, field_size=F, feature_size=V, embedding_size=K, deep_layers=(32, 16), dropout=(0.8, 0.8), cross_layers=2,
                    l2_reg=1e-3, learning_rate=1e-2, optimizer="Adam")
    params = O.init_params(ocfg, seed=2, scale=0.05)
    runs = []
    for p in (1, 1, period):        # (classic twice: what two runs of the SAME schedule differ by -- float atomics in the table step, amplified by Adam)
        eng = engine(model, p, V, B, params=params, keep=(0.8, 0.8))
        losses = []
        for step in range(30):
            b = B if step != 17 else 77
            ids, vals, labels = O.synth_batch(b, F, V, seed=900 + step)
            want = step in (0, 11, 29)
            losses.append(eng.train_step(*dev_batch(ids, vals, labels, dev), want_loss=want))
            if step == 20:
                pr = torch.empty(b, device=dev)
                eng.predict(*dev_batch(ids, vals, labels, dev)[:2], pr)
                losses.append(float(pr.sum()))
            if step == 23:
                losses.append(float(np.abs(eng.get_param("emb")).sum()))
        runs.append((losses, state_of(eng), eng.global_step))
        eng.close()
    assert runs[0][2] == runs[1][2] == runs[2][2] == 30
    noise_l = max([abs(a - b_) / max(1.0, abs(a)) for a, b_ in zip(runs[0][0], runs[1][0]) if a is not None] + [0.0])
    noise_v = max(float(np.abs(v - runs[1][1][k]).max()) for k, v in runs[0][1].items())
    for a, b_ in zip(runs[0][0], runs[2][0]):
        assert (a is None) == (b_ is None)
        if a is not None:
            assert abs(a - b_) <= max(3e-6, 4 * noise_l) * max(1.0, abs(a)), (a, b_, noise_l)
    for k, v in runs[0][1].items():
        assert np.abs(v - runs[2][1][k]).max() <= max(3e-6, 4 * noise_v), (k, noise_v)


def test_lagging_rows_match_the_oracle(dev):
    """12 steps with no loss read at all, then every variable against the oracle's dense Adam over all rows"""
    F, V, B, K = 39, 6000, 128, 8
    ocfg = O.Config(model="deepfm", field_size=F, feature_size=V, embedding_size=K, deep_layers=(32, 16), dropout=(1.0, 1.0), l2_reg=1e-3,
                    learning_rate=1e-2, optimizer="Adam")
    params = O.init_params(ocfg, seed=4, scale=0.05)
    eng = engine("deepfm", 8, V, B, params=params)
    oopt = O.Optimizer(ocfg, params)
    for step in range(12):
        ids, vals, labels = O.synth_batch(B, F, V, seed=950 + step)
        O.train_step(ocfg, params, oopt, ids, vals, labels)
        eng.train_step(*dev_batch(ids, vals, labels, dev), want_loss=False)
    got = eng.get_params()
    for name, ref in params.items():
        assert np.abs(got[name] - ref.numpy()).max() <= 2e-6, name
    for name in ("emb", "linear"):
        assert np.abs(eng.get_slot(name, 0) - oopt.slots[name]["m"].numpy()).max() <= 2e-6, name
        assert np.abs(eng.get_slot(name, 1) - oopt.slots[name]["v"].numpy()).max() <= 2e-6, name
    # a loss read after lagging steps: the l2 term covers the whole table as of the step
    ids, vals, labels = O.synth_batch(B, F, V, seed=990)
    ref_loss, _ = O.train_step(ocfg, params, oopt, ids, vals, labels)
    loss = eng.train_step(*dev_batch(ids, vals, labels, dev))
    assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss))
    eng.close()


def test_restored_global_step_and_written_parameters(dev):
    """set_global_step and set_param in the middle of training: rows are stamped relative to global_step and a written table is
    taken as of the present.  (At step 300 with five steps of history Adam moves every element ~0.03 per step -- lr_t is no longer
    bias-corrected down while v is still tiny -- and rounding noise grows with it: the yardstick is what two CLASSIC runs differ
    by; a row advanced with a neighbouring step's lr_t would be off by ~2e-5, a missed step by ~3e-2.)"""
    F, V, B, K = 39, 5000, 96, 8
    ocfg = O.Config(model="deepfm", field_size=F, feature_size=V, embedding_size=K, deep_layers=(16,), dropout=(1.0,), l2_reg=1e-3,
                    learning_rate=1e-2, optimizer="Adam")
    params = O.init_params(ocfg, seed=6, scale=0.05)
    runs = []
    for p in (1, 1, 6):
        eng = engine("deepfm", p, V, B, layers=(16,), params=params)
        for step in range(14):
            ids, vals, labels = O.synth_batch(B, F, V, seed=1000 + step)
            eng.train_step(*dev_batch(ids, vals, labels, dev), want_loss=False)
            if step == 4:
                eng.global_step = 300                    # a restored checkpoint's step (255 < 300: the stamps wrap)
            if step == 9:
                eng.set_param("linear", np.full((V,), 0.01, np.float32))
        runs.append(state_of(eng))
        assert eng.global_step == 309
        eng.close()
    noise = max(float(np.abs(v - runs[1][k]).max()) for k, v in runs[0].items())
    worst = max(float(np.abs(v - runs[2][k]).max()) for k, v in runs[0].items())
    print("classic vs classic %.3g, classic vs lagging %.3g" % (noise, worst))
    assert worst <= max(3e-6, 3 * noise), (worst, noise)


@pytest.mark.parametrize("model", ["deepfm", "dcn"])
def test_announced_batches(model, dev):
    """The input pipeline's hint (dctr_prefetch_ids after every step, batches in the engine's input slots): the next batch's ids are
    grouped during the step in flight, into the alternate grouping state.  Same results as the classic sweep without any hint; a step
    in the middle without a hint, one with a hint that is not honoured (another batch trained), a loss read and a short batch included.
    (Round 4 also tried advancing the announced batch's lagging rows ahead of its step -- profiles/r04_preadvance.txt: slower in every
    placement, and removed.)"""
    from tf_repos_amd import capi
    F, V, B, K = 39, 30000, 256, 8
    ocfg = O.Config(model=model, field_size=F, feature_size=V, embedding_size=K, deep_layers=(32, 16), dropout=(0.8, 0.8), cross_layers=2,
                    l2_reg=1e-3, learning_rate=1e-2, optimizer="Adam")
    params = O.init_params(ocfg, seed=8, scale=0.05)
    nb = capi.INPUT_SLOTS
    host = [O.synth_batch(B if i != 5 else 100, F, V, seed=1200 + i) for i in range(nb)]
    runs = []
    for period, hint in ((1, False), (1, False), (8, True)):
        eng = engine(model, period, V, B, params=params, keep=(0.8, 0.8))
        slots = []
        for i, (ids, vals, labels) in enumerate(host):
            si, sv, sl = eng.input_slot(i)
            b = len(labels)
            si[:b].copy_(torch.from_numpy(ids)); sv[:b].copy_(torch.from_numpy(vals)); sl[:b].copy_(torch.from_numpy(labels))
            slots.append((si[:b], sv[:b], sl[:b]))
        order = [s % nb for s in range(26)]
        losses = []
        for s, i in enumerate(order):
            losses.append(eng.train_step(*slots[i], want_loss=s in (0, 13, 25)))
            if hint and s + 1 < len(order) and s != 9:          # (step 10 comes unannounced)
                nxt = order[s + 1] if s != 17 else (order[s + 1] + 3) % nb      # (after step 17 the WRONG batch is announced)
                eng.prefetch_ids(slots[nxt][0])
        runs.append((losses, state_of(eng)))
        eng.close()
    noise = max(float(np.abs(v - runs[1][1][k]).max()) for k, v in runs[0][1].items())
    for a, b_ in zip(runs[0][0], runs[2][0]):
        assert (a is None) == (b_ is None)
        if a is not None:
            assert abs(a - b_) <= 1e-5 * max(1.0, abs(a)), (a, b_)
    for k, v in runs[0][1].items():
        assert np.abs(v - runs[2][1][k]).max() <= max(3e-6, 4 * noise), (k, noise)


def test_period_is_validated(dev):
    with pytest.raises(errors.InvalidArgumentError):
        engine("deepfm", 99, 1000, 16)


@pytest.mark.parametrize("model,att", [("din", ()), ("esmm", ()), ("din", (16,))])
def test_csr_models_lag_like_the_classic_sweep(model, att, dev):
    """DIN (sum and attention pooling) and ESMM over CSR batches (DIN.py:143-222, DeepCvrMTL.py:153-225): every entry's row is
    advanced to step t-1 when the multi-hot lookup (and the attention units' input) reads it; 14 steps, loss read at the ends and
    once in the middle, a predict in between."""
    from oracle import multihot_oracle as M
    B, Fc, V, K = 64, 6, 3000, 8
    ocfg = M.Config(model=model, field_size=Fc, feature_size=V, embedding_size=K, deep_layers=(32, 16), dropout=(0.8, 0.8), l2_reg=1e-3,
                    learning_rate=1e-2, optimizer="Adam", ctr_task_wgt=0.4, attention_layers=att)
    params = M.init_params(ocfg, seed=4, scale=0.2 if att else 0.05)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    runs = []
    for period in (1, 1, 6):       # (classic twice: the yardstick is what two runs of the same schedule differ by)
        eng = Engine(EngineConfig(model=model, field_size=ocfg.n_slots, feature_size=V, embedding_size=K, deep_layers=(32, 16), dropout=(0.8, 0.8),
                                  l2_reg=1e-3, learning_rate=1e-2, optimizer="Adam", max_batch=B, max_entries=B * (ocfg.n_slots + 40),
                                  ctr_task_wgt=0.4, attention_layers=att or (256,), att_pairs=[(Fc + i, Fc + 4 + i) for i in range(4)] if att else (),
                                  seed=9, table_sweep_period=period))
        eng.set_params(params)
        out = []
        for step in range(14):
            batch = M.synth_batch(ocfg, B, seed=1100 + step)
            off, ids, wts = M.slot_csr(ocfg, batch)
            out.append(eng.train_step_csr(t(off), t(ids), t(wts), t(batch["y"]), t(batch["z"]) if model == "esmm" else None,
                                          want_loss=step in (0, 6, 13)))
            if step == 9:
                p = torch.empty(B, device=dev)
                eng.predict_csr(t(off), t(ids), t(wts), B, p)
                out.append(float(p.sum()))
        st = dict(eng.get_params())
        st["emb/m"], st["emb/v"] = eng.get_slot("emb", 0), eng.get_slot("emb", 1)
        runs.append((out, st))
        eng.close()
    noise = max(float(np.abs(v - runs[1][1][k]).max()) for k, v in runs[0][1].items())
    for a, b in zip(runs[0][0], runs[2][0]):
        assert (a is None) == (b is None)
        if a is not None:
            assert abs(a - b) <= 1e-5 * max(1.0, abs(a)), (a, b)
    for k, v in runs[0][1].items():
        assert np.abs(v - runs[2][1][k]).max() <= max(3e-6, 4 * noise), (k, noise)      # (a missed or doubled step: ~1e-2)
