"""GPU parity of the CSR (multi-hot) models -- DIN with field-wise sum pooling (DIN.py:143-148,179-222) and ESMM
(DeepCvrMTL.py:153-225) -- through the C ABI (dctr_train_step_csr / dctr_predict_csr) against oracle/multihot_oracle.py on
identical seeded batches and injected weights.  Tolerances as for the fixed-field models: logits 1e-4, probabilities 1e-5,
loss 1e-5 relative, parameters after the steps 2e-6 absolute."""
import numpy as np
import pytest
import torch

from oracle import multihot_oracle as M
from tf_repos_amd import errors
from tf_repos_amd.engine import Engine, EngineConfig

pytestmark = pytest.mark.gpu


def make_pair(model, B, Fc=6, V=800, K=8, layers=(32, 16), opt="Adam", lr=1e-2, l2=1e-3, table_mode="dense_exact", wgt=0.5):
    ocfg = M.Config(model=model, field_size=Fc, feature_size=V, embedding_size=K, deep_layers=layers, dropout=(1.0,) * len(layers),
                    l2_reg=l2, learning_rate=lr, optimizer=opt, ctr_task_wgt=wgt)
    ecfg = EngineConfig(model=model, field_size=ocfg.n_slots, feature_size=V, embedding_size=K, deep_layers=layers,
                        dropout=(1.0,) * len(layers), l2_reg=l2, learning_rate=lr, optimizer=opt, table_mode=table_mode, max_batch=B,
                        max_entries=B * (ocfg.n_slots + 40), ctr_task_wgt=wgt)
    params = M.init_params(ocfg, seed=3)
    eng = Engine(ecfg)
    eng.set_params(params)
    return ocfg, params, eng


def dev_csr(ocfg, batch, dev):
    off, ids, wts = M.slot_csr(ocfg, batch)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t(off), t(ids), t(wts), t(batch["y"]), t(batch["z"])


@pytest.mark.parametrize("model", ["din", "esmm"])
@pytest.mark.parametrize("K", [4, 16])
def test_forward_matches_oracle(model, K, dev):
    B = 50
    ocfg, params, eng = make_pair(model, B, K=K)
    batch = M.synth_batch(ocfg, B, seed=5)
    ref = M.forward(ocfg, params, batch)
    off, ids, wts, _, _ = dev_csr(ocfg, batch, dev)
    o = [torch.empty(B, device=dev) for _ in range(3)]
    eng.predict_csr(off, ids, wts, B, *o)
    torch.cuda.synchronize()
    x = eng.debug_tensor("x_in")[:, :ocfg.n_slots * K].numpy()
    np.testing.assert_allclose(x, ref["x"].numpy(), rtol=0, atol=1e-6)          # gather + weighted segment sums
    if model == "din":
        assert np.abs(o[1].cpu().numpy() - ref["y"].numpy()).max() <= 1e-4      # logits
        assert np.abs(o[0].cpu().numpy() - ref["prob"].numpy()).max() <= 1e-5
    else:
        for got, key in zip(o, ("pctr", "pcvr", "pctcvr")):
            assert np.abs(got.cpu().numpy() - ref[key].numpy()).max() <= 1e-5, key
    eng.check_ids()
    eng.close()


@pytest.mark.parametrize("model", ["din", "esmm"])
@pytest.mark.parametrize("opt", ["Adam", "Adagrad", "Momentum", "ftrl"])
def test_train_steps_match_oracle(model, opt, dev):
    B = 64
    lr = {"Adam": 1e-2, "Adagrad": 1e-2, "Momentum": 1e-2, "ftrl": 5e-2}[opt]
    ocfg, params, eng = make_pair(model, B, opt=opt, lr=lr, wgt=0.3)
    oopt = M.Optimizer(ocfg, params)
    for step in range(3):
        batch = M.synth_batch(ocfg, B, seed=40 + step)
        ref_loss, _ = M.train_step(ocfg, params, oopt, batch)
        off, ids, wts, y, z = dev_csr(ocfg, batch, dev)
        loss = eng.train_step_csr(off, ids, wts, y, z if model == "esmm" else None)
        assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (step, loss, ref_loss)
    got = eng.get_params()
    for name, ref in params.items():
        diff = np.abs(got[name] - ref.numpy()).max()
        assert diff <= 2e-6, (name, diff)
    assert eng.global_step == 3
    eng.close()


def test_ragged_batches_empty_slots_and_no_weights(dev):
    """A short last batch, slots without entries (zeros, no gradient) and weights=None (all ones)."""
    B = 33
    ocfg, params, eng = make_pair("din", 64, opt="Adagrad")
    oopt = M.Optimizer(ocfg, params)
    batch = M.synth_batch(ocfg, B, seed=9, max_len=3)
    for n in M.MULTI_W:                                   # all weights one: the engine may then be given weights=None
        off, ids, _ = batch[n]
        batch[n] = (off, ids, np.ones(len(ids), np.float32))
    assert any((np.diff(batch[n][0]) == 0).any() for n in M.MULTI_W + M.MULTI_NW)
    ref_loss, _ = M.train_step(ocfg, params, oopt, batch)
    off, ids, _, y, _ = dev_csr(ocfg, batch, dev)
    loss = eng.train_step_csr(off, ids, None, y)
    assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss))
    got = eng.get_params()
    for name, ref in params.items():
        assert np.abs(got[name] - ref.numpy()).max() <= 2e-6, name
    eng.close()


def test_errors(dev):
    ocfg, params, eng = make_pair("din", 16)
    batch = M.synth_batch(ocfg, 16, seed=1)
    off, ids, wts, y, z = dev_csr(ocfg, batch, dev)
    with pytest.raises(errors.InvalidArgumentError):       # a fixed-field call on a CSR handle
        eng.train_step(ids[:16 * ocfg.n_slots].reshape(16, -1), wts[:16 * ocfg.n_slots].reshape(16, -1), y)
    bad = ids.clone()
    bad[3] = ocfg.feature_size + 7                          # out-of-range id: TF's gather raises InvalidArgumentError
    eng.predict_csr(off, bad, wts, 16, torch.empty(16, device=dev))
    with pytest.raises(errors.InvalidArgumentError):
        eng.check_ids()
    eng.close()
    ocfg, params, eng = make_pair("esmm", 16)
    with pytest.raises(errors.InvalidArgumentError):       # ESMM needs both labels
        eng.train_step_csr(off, ids, wts, y, None)
    eng.close()


def test_large_batch_with_hot_ids_uses_chunked_grouping(dev):
    """nnz > 8192 with a Zipf head: the id grouping aggregates 2048-entry chunks in an LDS hash table (group.hip) -- results must
    still match the oracle's dense segment sums."""
    B = 600
    ocfg, params, eng = make_pair("esmm", B, V=3000, opt="Adagrad", lr=1e-2)
    ecap = B * (ocfg.n_slots + 5 * 41)
    eng.close()
    ecfg = EngineConfig(model="esmm", field_size=ocfg.n_slots, feature_size=3000, embedding_size=8, deep_layers=(32, 16), dropout=(1.0, 1.0),
                        l2_reg=1e-3, learning_rate=1e-2, optimizer="Adagrad", max_batch=B, max_entries=ecap, ctr_task_wgt=0.5)
    eng = Engine(ecfg)
    eng.set_params(params)
    oopt = M.Optimizer(ocfg, params)
    for step in range(2):
        batch = M.synth_batch(ocfg, B, seed=80 + step, max_len=40)
        off, ids, wts, y, z = dev_csr(ocfg, batch, dev)
        assert ids.shape[0] > 4 * 2048
        ref_loss, _ = M.train_step(ocfg, params, oopt, batch)
        loss = eng.train_step_csr(off, ids, wts, y, z)
        assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (step, loss, ref_loss)
    got = eng.get_params()
    for name, ref in params.items():
        diff = np.abs(got[name] - ref.numpy()).max()
        assert diff <= 5e-6, (name, diff)          # segment sums of up to ~10^4 entries in a different order than the oracle's
    eng.close()


def make_att_pair(B, att=(16,), Fc=6, V=800, K=8, layers=(32, 16), opt="Adam", lr=1e-2, l2=1e-3):
    ocfg = M.Config(model="din", field_size=Fc, feature_size=V, embedding_size=K, deep_layers=layers, dropout=(1.0,) * len(layers),
                    l2_reg=l2, learning_rate=lr, optimizer=opt, attention_layers=att)
    ecfg = EngineConfig(model="din", field_size=ocfg.n_slots, feature_size=V, embedding_size=K, deep_layers=layers,
                        dropout=(1.0,) * len(layers), l2_reg=l2, learning_rate=lr, optimizer=opt, max_batch=B,
                        max_entries=B * (ocfg.n_slots + 40), attention_layers=att, att_pairs=[(Fc + i, Fc + 4 + i) for i in range(4)])
    params = M.init_params(ocfg, seed=4, scale=0.2)       # larger weights: the attention scores must matter
    eng = Engine(ecfg)
    eng.set_params(params)
    return ocfg, params, eng


@pytest.mark.parametrize("K,att", [(4, (8,)), (8, (16, 8)), (16, (32,))])
def test_din_attention_pooling_forward(K, att, dev):
    """attention_unit of DIN.py:152-172 for the four (user list, ad) pairs of DIN.py:174-177, variables shared"""
    B = 40
    ocfg, params, eng = make_att_pair(B, att=att, K=K)
    batch = M.synth_batch(ocfg, B, seed=6)
    batch["u_shop"][1][::5] = 0                             # id 0 = the padding id: masked out of the pooled sum (DIN.py:157)
    ref = M.forward(ocfg, params, batch)
    off, ids, wts, _, _ = dev_csr(ocfg, batch, dev)
    prob, logit = torch.empty(B, device=dev), torch.empty(B, device=dev)
    eng.predict_csr(off, ids, wts, B, prob, logit)
    torch.cuda.synchronize()
    x = eng.debug_tensor("x_in")[:, :ocfg.n_slots * K].numpy()
    np.testing.assert_allclose(x, ref["x"].numpy(), rtol=0, atol=2e-6)
    assert np.abs(logit.cpu().numpy() - ref["y"].numpy()).max() <= 1e-4
    eng.close()


@pytest.mark.parametrize("opt", ["Adam", "Adagrad", "Momentum", "ftrl"])
def test_din_attention_pooling_train_steps(opt, dev):
    B = 48
    lr = {"Adam": 1e-2, "Adagrad": 1e-2, "Momentum": 1e-2, "ftrl": 5e-2}[opt]
    ocfg, params, eng = make_att_pair(B, att=(16, 8), opt=opt, lr=lr)
    oopt = M.Optimizer(ocfg, params)
    for step in range(3):
        batch = M.synth_batch(ocfg, B, seed=90 + step)
        batch["u_cat"][1][::7] = 0
        ref_loss, _ = M.train_step(ocfg, params, oopt, batch)
        off, ids, wts, y, _ = dev_csr(ocfg, batch, dev)
        loss = eng.train_step_csr(off, ids, wts, y)
        assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (step, loss, ref_loss)
    got = eng.get_params()
    for name, ref in params.items():
        diff = np.abs(got[name] - ref.numpy()).max()
        assert diff <= 3e-6, (name, diff)
    eng.close()


@pytest.mark.parametrize("model", ["din", "esmm"])
@pytest.mark.parametrize("opt", ["Momentum", "Adagrad"])
def test_batch_norm_towers_match_oracle(model, opt, dev):
    """--batch_norm=True: batch_norm_layer after every hidden ReLU of the tower(s) (DIN.py:203-204, DeepCvrMTL.py:177-178,198-199):
    batch statistics + in-place moving averages in TRAIN, moving statistics in PREDICT.  (Adam is left out as for the other BN
    tests: the bias gradients are exactly zero in exact arithmetic and Adam amplifies their rounding noise.)"""
    B = 64
    layers = (32, 16)
    ocfg = M.Config(model=model, field_size=6, feature_size=800, embedding_size=8, deep_layers=layers, dropout=(1.0, 1.0), l2_reg=1e-3,
                    learning_rate=1e-2, optimizer=opt, ctr_task_wgt=0.4, batch_norm=True, batch_norm_decay=0.9)
    ecfg = EngineConfig(model=model, field_size=ocfg.n_slots, feature_size=800, embedding_size=8, deep_layers=layers, dropout=(1.0, 1.0),
                        l2_reg=1e-3, learning_rate=1e-2, optimizer=opt, max_batch=B, max_entries=B * (ocfg.n_slots + 40), ctr_task_wgt=0.4,
                        batch_norm=True, batch_norm_decay=0.9)
    params = M.init_params(ocfg, seed=5)
    eng = Engine(ecfg)
    eng.set_params(params)
    oopt = M.Optimizer(ocfg, params)
    for step in range(3):
        batch = M.synth_batch(ocfg, B, seed=120 + step)
        ref_loss, _ = M.train_step(ocfg, params, oopt, batch)
        off, ids, wts, y, z = dev_csr(ocfg, batch, dev)
        loss = eng.train_step_csr(off, ids, wts, y, z if model == "esmm" else None)
        assert abs(loss - ref_loss) <= 2e-5 * max(1.0, abs(ref_loss)), (step, loss, ref_loss)
    got = eng.get_params()
    for name, ref in params.items():
        diff = np.abs(got[name] - ref.numpy()).max()
        assert diff <= 2e-5, (name, diff)                 # tolerance of the other batch_norm tests
    batch = M.synth_batch(ocfg, B, seed=7)
    ref = M.forward(ocfg, params, batch)                   # PREDICT: moving statistics
    off, ids, wts, _, _ = dev_csr(ocfg, batch, dev)
    o = [torch.empty(B, device=dev) for _ in range(3)]
    eng.predict_csr(off, ids, wts, B, *o)
    torch.cuda.synchronize()
    key = "prob" if model == "din" else "pctr"
    assert np.abs(o[0].cpu().numpy() - ref[key].numpy()).max() <= 2e-5
    eng.close()
