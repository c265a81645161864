"""GPU parity: the HIP engine (through the C ABI) vs the CPU oracle on identical seeded batches and injected
weights.  Tolerances from BASELINE.json north_star: logits within 1e-4 (fp32); parameters after an optimizer
step 1e-6 abs."""
import numpy as np
import pytest
import torch

from oracle import deepctr_oracle as O
from tests.util import dev_batch, make_pair

pytestmark = pytest.mark.gpu

MODELS = ["deepfm", "fnn", "ipnn", "nfm", "dcn", "afm", "mvm"]


@pytest.mark.parametrize("model", MODELS + ["opnn"])
@pytest.mark.parametrize("K", [4, 8, 16, 32])
def test_forward_logits(model, K, dev):
    # (opnn: K < 16 takes the materialising kernels, K >= 16 forms the pair products inside the first layer's GEMM)
    F, V, B = (39, 5000, 96) if model != "opnn" else (10, 500, 32)
    ocfg, params, eng = make_pair(model, B=B, F=F, V=V, K=K, layers=(64, 32))
    ids, vals, labels = O.synth_batch(B, F, V, seed=7)
    ref = O.forward(ocfg, params, ids, vals)
    d_ids, d_vals, _ = dev_batch(ids, vals, labels, dev)
    prob = torch.empty(B, device=dev)
    logit = torch.empty(B, device=dev)
    eng.predict(d_ids, d_vals, prob, logit)
    torch.cuda.synchronize()
    assert np.abs(logit.cpu().numpy() - ref["y"].numpy()).max() <= 1e-4      # tolerance: 1e-4 abs on logits
    assert np.abs(prob.cpu().numpy() - ref["prob"].numpy()).max() <= 1e-5
    e = eng.debug_tensor("e")[:, :F * K].numpy().reshape(B, F, K)
    np.testing.assert_allclose(e, ref["e"].numpy(), rtol=0, atol=1e-7)
    eng.check_ids()
    eng.close()


@pytest.mark.parametrize("model", MODELS)
@pytest.mark.parametrize("opt", ["Adam", "Adagrad", "Momentum", "ftrl"])
def test_train_steps_match_oracle(model, opt, dev):
    F, V, B, K = 39, 3000, 128, 8
    lr = {"Adam": 1e-2, "Adagrad": 1e-2, "Momentum": 1e-2, "ftrl": 5e-2}[opt]
    ocfg, params, eng = make_pair(model, B=B, F=F, V=V, K=K, layers=(32, 16), opt=opt, lr=lr)
    oopt = O.Optimizer(ocfg, params)
    for step in range(3):
        ids, vals, labels = O.synth_batch(B, F, V, seed=100 + step)
        ref_loss, _ = O.train_step(ocfg, params, oopt, ids, vals, labels)
        d = dev_batch(ids, vals, labels, dev)
        loss = eng.train_step(*d)
        assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (step, loss, ref_loss)
    got = eng.get_params()
    for name, ref in params.items():
        diff = np.abs(got[name] - ref.numpy()).max()
        assert diff <= 2e-6, (name, diff)       # tolerance: per-parameter 1e-6-class abs after 3 steps
    assert eng.global_step == 3
    eng.close()


@pytest.mark.parametrize("K,H,B", [(16, 256, 96), (16, 200, 77), (32, 64, 130), (16, 320, 64), (64, 128, 40), (32, 256, 33)])
def test_outer_pnn_fused_first_layer(K, H, B, dev):
    """Outer-PNN with K >= 16: the [B, P K K] product tensor is never formed (gemm_dr.h DR_AGEN_*); forward, weight gradient and
    dL/de against the oracle's materialised einsum (PNN.py:139-167), ragged batches, every tile width of the kernel."""
    F, V = 10, 700
    ocfg, params, eng = make_pair("opnn", B=B, F=F, V=V, K=K, layers=(H, 32), opt="Adam", lr=1e-3, l2=1e-4, scale=0.05)
    oopt = O.Optimizer(ocfg, params)
    for step in range(2):
        ids, vals, labels = O.synth_batch(B, F, V, seed=300 + step)
        d = dev_batch(ids, vals, labels, dev)
        if step == 0:
            ref = O.forward(ocfg, params, ids, vals)
            logit = torch.empty(B, device=dev)
            eng.predict(d[0], d[1], torch.empty(B, device=dev), logit)
            assert np.abs(logit.cpu().numpy() - ref["y"].numpy()).max() <= 1e-4
        ref_loss, _ = O.train_step(ocfg, params, oopt, ids, vals, labels)
        loss = eng.train_step(*d)
        assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (step, loss, ref_loss)
    got = eng.get_params()
    for name, ref in params.items():
        # (5e-6: dL/de is summed by float atomics -- 12 waves of a block and the pair-split blocks meet in the same rows in no fixed order)
        assert np.abs(got[name] - ref.numpy()).max() <= 5e-6, name
    eng.close()


@pytest.mark.parametrize("K,H,B,F", [(16, 64, 50, 6), (32, 256, 70, 5)])
def test_outer_pnn_first_layer_ops_through_the_c_abi(K, H, B, F, dev):
    """dctr_pnn_outer_fc_{fwd,bwd_weights,bwd_data} against the materialised form of PNN.py:154-167 + fully_connected in fp64."""
    from tf_repos_amd import capi
    L = capi.lib()
    g = torch.Generator().manual_seed(3)
    P = F * (F - 1) // 2
    e = torch.randn(B, F * K, generator=g) * 0.3
    w = torch.randn(F * K + P * K * K, H, generator=g) * 0.05
    b = torch.randn(H, generator=g) * 0.1
    dy = torch.randn(B, H, generator=g) * 0.1
    e3 = e.double().view(B, F, K)
    ops = [torch.einsum("ba,bc->bac", e3[:, i], e3[:, j]).reshape(B, K * K) for i in range(F) for j in range(i + 1, F)]     # PNN.py:142-146,161-166
    x = torch.cat([e.double()] + ops, dim=1).requires_grad_(True)
    wd = w.double().requires_grad_(True)
    y_ref = torch.relu(x @ wd + b.double())
    y_ref.backward(dy.double() * (y_ref > 0))       # (the C ABI takes the gradient at the pre-activation, like dctr_fc_bwd_*)
    pre = dy.double() * (y_ref > 0).double()
    # dE through both row groups of W: chain rule of the products by hand
    dx = x.grad
    dE = dx[:, :F * K].clone().view(B, F, K)
    col = F * K
    for i in range(F):
        for j in range(i + 1, F):
            d = dx[:, col:col + K * K].view(B, K, K)
            dE[:, i] += torch.einsum("bac,bc->ba", d, e3[:, j])
            dE[:, j] += torch.einsum("bac,ba->bc", d, e3[:, i])
            col += K * K
    d_e, d_w, d_b, d_pre = e.to(dev), w.to(dev), b.to(dev), pre.float().to(dev)
    y = torch.empty(B, H, device=dev)
    nws = L.dctr_pnn_outer_fc_workspace_bytes(B, H)
    ws = torch.empty(nws // 4, device=dev)
    st = capi.current_stream()
    capi.check(L.dctr_pnn_outer_fc_fwd(capi.ptr(d_e), F * K, B, F, K, capi.ptr(d_w), capi.ptr(d_b), capi.ptr(y), H, H, 1, 1.0, 0, capi.ptr(ws), nws, st))
    assert (y.cpu().double() - y_ref.detach()).abs().max() <= 1e-5
    dw, db = torch.empty_like(d_w), torch.empty(H, device=dev)
    capi.check(L.dctr_pnn_outer_fc_bwd_weights(capi.ptr(d_e), F * K, B, F, K, capi.ptr(d_pre), H, H, capi.ptr(dw), capi.ptr(db), st))
    assert (dw.cpu().double() - wd.grad).abs().max() <= 1e-5
    assert (db.cpu().double() - pre.sum(0)).abs().max() <= 1e-5
    dEg = torch.full((B, F * K), 7.0, device=dev)           # overwritten
    capi.check(L.dctr_pnn_outer_fc_bwd_data(capi.ptr(d_e), F * K, B, F, K, capi.ptr(d_pre), H, H, capi.ptr(d_w), capi.ptr(dEg), F * K, st))
    assert (dEg.cpu().double() - dE.reshape(B, F * K)).abs().max() <= 1e-5


@pytest.mark.parametrize("K,A,B,F,keep", [(16, 32, 96, 39, (1.0, 1.0)), (16, 32, 64, 12, (0.7, 0.6)), (256, 128, 24, 39, (0.5, 0.5)), (8, 16, 50, 10, (1.0, 1.0)),
                                          (256, 256, 128, 39, (0.5, 0.5)), (256, 128, 130, 39, (1.0, 1.0))])
def test_afm_interaction_ops_through_the_c_abi(K, A, B, F, keep, dev):
    """dctr_afm_fwd / dctr_afm_bwd (AFM.py:127-158: pair products, attention network, softmax over the pairs, both dropouts, pooling)
    against the same lines written out in fp64 with autograd; the attention variables' gradients through dctr_param_grad_get.
    K = 16 takes the fused attention kernels, K = 256 / A = 128 (run.sh:18) the layer-by-layer path with the products carrying
    attention_out, K = 8 the small-row forms; K = 256 at B = 128 / 130 (95 k pair rows: run.sh:18's own batch) the tall split-precision products in the
    default gemm mode (csrc/gemm_ts.h: neither the pair tensor nor the attention layer's output is written) and the f32 kernels in the exact one."""
    from tf_repos_amd import capi
    from tf_repos_amd.engine import Engine, EngineConfig
    P = F * (F - 1) // 2
    g = torch.Generator().manual_seed(11)
    e = torch.randn(B, F, K, generator=g) * 0.3
    W = torch.randn(K, A, generator=g) * 0.2
    b = torch.randn(A, generator=g) * 0.1
    wo = torch.randn(A, 1, generator=g) * 0.3
    bo = torch.randn(1, generator=g) * 0.1
    dy = torch.randn(B, K, generator=g) * 0.1
    eng = Engine(EngineConfig(model="afm", field_size=F, feature_size=100, embedding_size=K, deep_layers=(1,), attention_layers=(A,), dropout=keep,
                              l2_reg=0.0, learning_rate=1e-3, optimizer="Adam", max_batch=B, seed=5))
    for name, v in (("att_mlp0/weights", W), ("att_mlp0/biases", b), ("attention_out/weights", wo), ("attention_out/biases", bo)):
        eng.set_param(name, v.numpy())
    train = min(keep) < 1.0
    # (the op draws its masks at the handle's CURRENT step state: a fresh handle stands at global_step 0, seed_t = seed)
    m_att = torch.from_numpy(eng.dropout_mask(capi.SITE_AFM_ATT, (B, P, 1), keep[0], step=0).astype(np.float64)) if train else torch.ones(B, P, 1, dtype=torch.float64)
    m_emb = torch.from_numpy(eng.dropout_mask(capi.SITE_AFM_YEMB, (B, K), keep[1], step=0).astype(np.float64)) if train else torch.ones(B, K, dtype=torch.float64)
    if train:
        assert 0.05 < float(m_att.mean()) < 0.95 and 0.05 < float(m_emb.mean()) < 0.95
    # fp64 reference with autograd
    e64 = e.double().requires_grad_(True)
    prm = [t.double().requires_grad_(True) for t in (W, b, wo, bo)]
    row = [i for i in range(F - 1) for _ in range(i + 1, F)]
    col = [j for i in range(F - 1) for j in range(i + 1, F)]
    pp = e64[:, row, :] * e64[:, col, :]                                        # AFM.py:134-138
    z = pp.reshape(-1, K) @ prm[0] + prm[1]
    ah = torch.relu(z)                                                          # AFM.py:142-145
    sc = (ah @ prm[2] + prm[3]).reshape(B, P, 1)                                # AFM.py:147
    sc.retain_grad()
    soft = torch.softmax(sc, dim=1)                                             # AFM.py:151
    a_d = soft * m_att / keep[0]                                                # AFM.py:152-153
    y = (a_d * pp).sum(1) * m_emb / keep[1]                                     # AFM.py:156-158
    y.backward(dy.double())
    y_gpu, att = eng.afm_fwd(e.reshape(B, F * K).to(dev), train=train, want_att=True)
    assert (y_gpu.cpu().double() - y.detach()).abs().max() <= 2e-6 * max(1.0, float(y.detach().abs().max()))
    assert (att.cpu().double() - soft.detach().reshape(B, P)).abs().max() <= 1e-6         # (the softmax weights, before their dropout)
    dE = eng.afm_bwd(dy.to(dev))
    assert (dE.cpu().double() - e64.grad.reshape(B, F * K)).abs().max() <= 2e-6 * max(1.0, float(e64.grad.abs().max()))
    # ReLU decisions at |z| below fp32's resolution of z can fall either way in ANY fp32 evaluation (24 M of them at K = A = 256, B = 128: three
    # within 1e-7 of zero, one within 3e-8); one that does moves dW[:, a] by d sc[r] w_o[a] pp[r, :] and db[a] by d sc[r] w_o[a].  Budget for exactly
    # those (tools/afm_relu_flip_diag.py: the one column with such a z carries 4.7e-7 of error in split mode, every other column <= 2e-8).
    near = (z.detach().abs() < 2e-7).double()
    dsc = sc.grad.reshape(-1, 1).abs()
    slack = {"att_mlp0/weights": ((pp.detach().reshape(-1, K).abs() * dsc).t() @ near) * prm[2].detach().abs().reshape(1, -1),
             "att_mlp0/biases": ((dsc * near).sum(0)) * prm[2].detach().abs().reshape(-1)}
    for name, t in zip(("att_mlp0/weights", "att_mlp0/biases", "attention_out/weights", "attention_out/biases"), prm):
        got = eng.get_grad(name).astype(np.float64).reshape(t.shape)
        scale = max(1e-3, float(t.grad.abs().max()))
        # (attention_out's bias has gradient exactly 0 -- the softmax is shift-invariant -- and fp32 leaves ~1e-8 of rounding there)
        tol = max(3e-6 * scale, 1e-7) + (slack[name].numpy().reshape(t.shape) if name in slack else 0.0)
        err = np.abs(got - t.grad.numpy())
        assert (err <= tol).all(), (name, float(err.max()), scale, int(near.sum()), float(np.max(tol)))
    eng.close()


@pytest.mark.parametrize("att", [(16, 8), (24, 12, 8)])
def test_afm_multi_layer_attention(att, dev):
    """AFM.py:143-145 loops over --attention_layers: more than one width runs layer by layer over the B*P pair rows."""
    F, V, B, K = 12, 800, 48, 8
    ocfg, params, eng = make_pair("afm", B=B, F=F, V=V, K=K, att=att, opt="Adam", lr=1e-2, l2=1e-3)
    assert all("att_mlp%d/weights" % i in eng.param_shapes for i in range(len(att)))
    oopt = O.Optimizer(ocfg, params)
    for step in range(2):
        ids, vals, labels = O.synth_batch(B, F, V, seed=500 + step)
        d = dev_batch(ids, vals, labels, dev)
        if step == 0:
            ref = O.forward(ocfg, params, ids, vals)
            logit = torch.empty(B, device=dev)
            eng.predict(d[0], d[1], torch.empty(B, device=dev), logit)
            assert np.abs(logit.cpu().numpy() - ref["y"].numpy()).max() <= 1e-4
        ref_loss, _ = O.train_step(ocfg, params, oopt, ids, vals, labels)
        loss = eng.train_step(*d)
        assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (step, loss, ref_loss)
    got = eng.get_params()
    for name, ref in params.items():
        assert np.abs(got[name] - ref.numpy()).max() <= 2e-6, name
    eng.close()


def test_touched_rows_mode_only_updates_batch_rows(dev):
    F, V, B, K = 39, 3000, 64, 8
    ocfg, params, eng = make_pair("deepfm", B=B, F=F, V=V, K=K, table_mode="touched_rows")
    ids, vals, labels = O.synth_batch(B, F, V, seed=5)
    before = eng.get_param("emb")
    eng.train_step(*dev_batch(ids, vals, labels, dev))
    after = eng.get_param("emb")
    touched = np.zeros(V, bool)
    touched[np.unique(ids)] = True
    assert np.array_equal(before[~touched], after[~touched])
    assert np.abs(before[touched] - after[touched]).max() > 0
    eng.close()


def test_out_of_range_id_raises(dev):
    from tf_repos_amd import errors
    F, V, B, K = 39, 1000, 32, 8
    ocfg, params, eng = make_pair("deepfm", B=B, F=F, V=V, K=K)
    ids, vals, labels = O.synth_batch(B, F, V, seed=5)
    ids[3, 20] = V + 5
    d_ids, d_vals, _ = dev_batch(ids, vals, labels, dev)
    eng.predict(d_ids, d_vals, torch.empty(B, device=dev), None)
    with pytest.raises(errors.InvalidArgumentError):
        eng.check_ids()
    eng.close()


def test_graph_and_eager_agree_and_ragged_last_batch(dev):
    F, V, K = 39, 3000, 8
    res = []
    for use_graph in (True, False):
        ocfg, params, eng = make_pair("deepfm", B=128, F=F, V=V, K=K, use_graph=use_graph)
        for B in (128, 37, 128):       # a short last batch (DeepFM.py:92 batches may be short)
            ids, vals, labels = O.synth_batch(B, F, V, seed=B)
            eng.train_step(*dev_batch(ids, vals, labels, dev))
        res.append(eng.get_params())
        eng.close()
    for k in res[0]:
        assert np.abs(res[0][k] - res[1][k]).max() <= 1e-6, k


@pytest.mark.parametrize("model,opt", [("deepfm", "Momentum"), ("deepfm", "Adagrad"), ("nfm", "Momentum"), ("dcn", "Momentum"), ("ipnn", "Adagrad")])
def test_batch_norm_matches_oracle(model, opt, dev):
    """--batch_norm=True (run.sh:17 uses it for NFM): contrib batch_norm after each hidden ReLU (DeepFM.py:159-160,231-235):
    batch statistics + moving-average update in TRAIN, moving statistics in EVAL/PREDICT.
    Not Adam here: BN makes the loss invariant to a shift of its input, so the gradient of a bias whose units are active for the
    whole batch is EXACTLY zero in exact arithmetic and ~1e-9 rounding noise in f32 -- Adam divides by sqrt(v)+1e-8 and turns
    that noise into O(lr/100) steps on either side (the f32 oracle differs from its own f64 shadow the same way)."""
    from tf_repos_amd.engine import Engine, EngineConfig
    F, V, B, K = 39, 3000, 128, 8
    ocfg = O.Config(model=model, field_size=F, feature_size=V, embedding_size=K, deep_layers=(32, 16), dropout=(1.0, 1.0), cross_layers=2,
                    l2_reg=1e-3, learning_rate=1e-2, optimizer=opt, batch_norm=True, batch_norm_decay=0.9)
    params = O.init_params(ocfg, seed=3, scale=0.05)
    rng = np.random.default_rng(0)
    for k in params:          # non-trivial BN parameters / statistics
        if k.endswith(("gamma", "moving_variance")):
            params[k] = torch.from_numpy(rng.uniform(0.5, 1.5, size=tuple(params[k].shape)).astype(np.float32))
        if k.endswith(("beta", "moving_mean")):
            params[k] = torch.from_numpy(rng.normal(0, 0.1, size=tuple(params[k].shape)).astype(np.float32))
    eng = Engine(EngineConfig(model=model, field_size=F, feature_size=V, embedding_size=K, deep_layers=(32, 16), dropout=(1.0, 1.0),
                              cross_layers=2, l2_reg=1e-3, learning_rate=1e-2, optimizer=opt, batch_norm=True, batch_norm_decay=0.9,
                              max_batch=B, use_graph=False))
    assert set(eng.param_shapes) == set(params)
    eng.set_params(params)
    ids, vals, labels = O.synth_batch(B, F, V, seed=5)
    d = dev_batch(ids, vals, labels, dev)
    logit = torch.empty(B, device=dev)
    prob = torch.empty(B, device=dev)
    eng.predict(d[0], d[1], prob, logit)                                   # inference: moving statistics
    ref = O.forward(ocfg, params, ids, vals, train=False)
    assert np.abs(logit.cpu().numpy() - ref["y"].numpy()).max() <= 1e-4
    oopt = O.Optimizer(ocfg, params)
    for step in range(3):
        ids, vals, labels = O.synth_batch(B, F, V, seed=200 + step)
        ref_loss, _ = O.train_step(ocfg, params, oopt, ids, vals, labels)
        loss = eng.train_step(*dev_batch(ids, vals, labels, dev))
        assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (step, loss, ref_loss)
    got = eng.get_params()
    for name, refv in params.items():
        assert np.abs(got[name] - refv.numpy()).max() <= 2e-5, name         # includes moving_mean / moving_variance; BN divides by the batch std, which amplifies f32 rounding ~10x
    eng.close()


def test_batch_norm_with_dropout_runs_and_freezes_moving_stats_under_ftrl(dev):
    from tf_repos_amd.engine import Engine, EngineConfig
    F, V, B, K = 39, 3000, 64, 8
    eng = Engine(EngineConfig(model="deepfm", field_size=F, feature_size=V, embedding_size=K, deep_layers=(32, 16), dropout=(0.5, 0.5),
                              l2_reg=1e-3, learning_rate=1e-2, optimizer="ftrl", batch_norm=True, batch_norm_decay=0.5, max_batch=B))
    rng = np.random.default_rng(1)
    for name, shp in eng.param_shapes.items():
        eng.set_param(name, np.ones(shp, np.float32) if name.endswith(("gamma", "moving_variance")) else rng.normal(0, 0.05, size=shp).astype(np.float32))
    ids, vals, labels = O.synth_batch(B, F, V, seed=9)
    loss = eng.train_step(*dev_batch(ids, vals, labels, dev))
    assert np.isfinite(loss)
    mv = eng.get_param("bn_0/moving_variance")
    mm = eng.get_param("bn_0/moving_mean")
    assert np.all(mv > 0.4) and np.all(mv < 1.0) and np.abs(mm).max() > 0         # moved by the moving average only, not zeroed by Ftrl
    eng.close()


@pytest.mark.parametrize("opt", ["Adam", "Adagrad"])
def test_deepmvm_product_layer_with_factors_near_one(opt, dev):
    """DeepMVM.py:144-150: x_mvm = prod_f (e_f + mvm_b_f).  With N(0, 0.05) weights the 39-factor product underflows and the
    layer is invisible; factors near 1 make its forward and backward matter."""
    F, V, B, K = 39, 3000, 128, 8
    ocfg, params, eng = make_pair("mvm", B=B, F=F, V=V, K=K, layers=(32, 16), opt=opt, lr=1e-2)
    rng = np.random.default_rng(2)
    params["mvm_b"] = torch.from_numpy((1.0 + rng.normal(0, 0.1, size=(F, K))).astype(np.float32))
    eng.set_params(params)
    ids, vals, labels = O.synth_batch(B, F, V, seed=31)
    ref = O.forward(ocfg, params, ids, vals)
    assert float(ref["x_mvm"].abs().mean()) > 1e-2                      # the product is alive
    d = dev_batch(ids, vals, labels, dev)
    logit = torch.empty(B, device=dev)
    eng.predict(d[0], d[1], torch.empty(B, device=dev), logit)
    assert np.abs(logit.cpu().numpy() - ref["y"].numpy()).max() <= 1e-4
    oopt = O.Optimizer(ocfg, params)
    for step in range(3):
        ids, vals, labels = O.synth_batch(B, F, V, seed=300 + step)
        ref_loss, _ = O.train_step(ocfg, params, oopt, ids, vals, labels)
        loss = eng.train_step(*dev_batch(ids, vals, labels, dev))
        assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss))
    got = eng.get_params()
    for name, refv in params.items():
        assert np.abs(got[name] - refv.numpy()).max() <= 5e-6, name
    eng.close()


@pytest.mark.parametrize("K,A", [(4, 16), (8, 48), (16, 100), (16, 256), (32, 64), (8, 300)])
def test_afm_attention_widths(K, A, dev):
    """The attention network runs fused over the pair rows for K in {4,8,16,32}, A <= 256 (hidden width padded to 32/64/128/256
    columns in LDS) and through the GEMM path otherwise ((8,300)); both must match AFM.py:141-166 with the same tolerances."""
    F, V, B = 13, 2000, 70          # B*P = 5460 pair rows: a ragged last 32-row tile
    ocfg, params, eng = make_pair("afm", B=B, F=F, V=V, K=K, layers=(1,), att=(A,), opt="Adagrad", lr=1e-2)
    oopt = O.Optimizer(ocfg, params)
    for step in range(2):
        ids, vals, labels = O.synth_batch(B, F, V, seed=300 + step)
        ref_loss, _ = O.train_step(ocfg, params, oopt, ids, vals, labels)
        loss = eng.train_step(*dev_batch(ids, vals, labels, dev))
        assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (step, loss, ref_loss)
    got = eng.get_params()
    for name, ref in params.items():
        diff = np.abs(got[name] - ref.numpy()).max()
        assert diff <= 2e-6, (name, diff)
    ids, vals, labels = O.synth_batch(B, F, V, seed=9)
    d_ids, d_vals, _ = dev_batch(ids, vals, labels, dev)
    logit = torch.empty(B, device=dev)
    eng.predict(d_ids, d_vals, torch.empty(B, device=dev), logit)
    torch.cuda.synchronize()
    assert np.abs(logit.cpu().numpy() - O.forward(ocfg, params, ids, vals)["y"].numpy()).max() <= 1e-4


@pytest.mark.parametrize("K,A,F,B", [(128, 128, 12, 1100), (256, 128, 9, 1900), (128, 256, 12, 1000)])
def test_afm_attention_out_inside_the_products(K, A, F, B, dev):
    """From 65536 pair rows on (and K, A in {128, 256}) the score dot comes out of the attention product's epilogue, and the
    backward never writes d ah = dsc (x) w_o . 1[ah > 0]: the input gradient gates its operand loads on ah's sign, the weight
    gradient does the same and its second column sums are attention_out's dW (AFM.py:142-148).  Ragged last tiles on purpose."""
    V = 3000
    assert B * F * (F - 1) // 2 >= 65536
    ocfg, params, eng = make_pair("afm", B=B, F=F, V=V, K=K, layers=(1,), att=(A,), opt="Adagrad", lr=1e-2, l2=1e-3)
    oopt = O.Optimizer(ocfg, params)
    for step in range(2):
        ids, vals, labels = O.synth_batch(B, F, V, seed=700 + step)
        ref_loss, _ = O.train_step(ocfg, params, oopt, ids, vals, labels)
        loss = eng.train_step(*dev_batch(ids, vals, labels, dev))
        assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (step, loss, ref_loss)
    got = eng.get_params()
    for name, ref in params.items():
        diff = np.abs(got[name] - ref.numpy()).max()
        assert diff <= 3e-6, (name, diff)
    # a smaller batch on the same engine falls back to the materialising passes (the slabs are laid out for the products)
    ids, vals, labels = O.synth_batch(200, F, V, seed=702)
    ref_loss, _ = O.train_step(ocfg, params, oopt, ids, vals, labels)
    loss = eng.train_step(*dev_batch(ids, vals, labels, dev))
    assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss))
    got = eng.get_params()
    for name, ref in params.items():
        diff = np.abs(got[name] - ref.numpy()).max()
        assert diff <= 4e-6, (name, diff)
    eng.close()


@pytest.mark.parametrize("K,A,F,B", [(128, 128, 12, 5001), (256, 256, 9, 7300), (256, 128, 10, 5900)])
def test_afm_tall_split_precision_products_in_the_step(K, A, F, B, dev):
    """From 65536 pair rows and 2.5 G multiply-adds per product on, a handle in the default (split) gemm mode runs the attention layer's three
    products on the bf16 matrix pipe (csrc/gemm_ts.h): forward + score with the pair rows e_i . e_j formed in the registers (the [B P, K]
    tensor of AFM.py:130-139 is never written) and the sign bits of the output beside it, gated input gradient from those bits, gated weight
    gradient from 256 partial slabs -- against the oracle at the tolerances of the exact path.  Then two smaller batches on the same handle:
    ~70000 rows (one block per row tile instead of looping blocks; at K = A = 128 below the work threshold: the forward and the input gradient
    go back to the f32 kernels while the weight gradient keeps the 256 slabs the handle declared -- most of them short or empty), and below
    65536 rows, where everything takes the materialising passes."""
    import os
    from tf_repos_amd import capi
    V = 3000
    P = F * (F - 1) // 2
    assert B * P >= 262144 and B * P * K * A >= 5e9
    ocfg, params, eng = make_pair("afm", B=B, F=F, V=V, K=K, layers=(1,), att=(A,), opt="Adagrad", lr=1e-2, l2=1e-3)
    oopt = O.Optimizer(ocfg, params)
    lib = capi.lib()
    n0 = lib.dctr_gemm_split_launches()
    for step in range(2):
        ids, vals, labels = O.synth_batch(B, F, V, seed=900 + step)
        ref_loss, _ = O.train_step(ocfg, params, oopt, ids, vals, labels)
        loss = eng.train_step(*dev_batch(ids, vals, labels, dev))
        assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (step, loss, ref_loss)
    if os.environ.get("DCTR_GEMM_MODE", "split") not in ("exact", "2"):
        assert lib.dctr_gemm_split_launches() - n0 >= 3, "the tall split-precision kernels did not run"      # (3 per step; a captured step enqueues them once)
    got = eng.get_params()
    for name, ref in params.items():
        diff = np.abs(got[name] - ref.numpy()).max()
        assert diff <= 3e-6, (name, diff)
    for b2, seed in ((max(70000 // P + 1, 300), 902), (150, 903)):
        ids, vals, labels = O.synth_batch(b2, F, V, seed=seed)
        ref_loss, _ = O.train_step(ocfg, params, oopt, ids, vals, labels)
        loss = eng.train_step(*dev_batch(ids, vals, labels, dev))
        assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (b2, loss, ref_loss)
        got = eng.get_params()
        for name, ref in params.items():
            diff = np.abs(got[name] - ref.numpy()).max()
            assert diff <= 4e-6, (name, b2, diff)
    # inference on the big batch (the forward alone: no backward follows the unwritten pair tensor)
    ids, vals, labels = O.synth_batch(B, F, V, seed=904)
    d = dev_batch(ids, vals, labels, dev)
    logit, prob = torch.empty(B, device=dev), torch.empty(B, device=dev)
    eng.predict(d[0], d[1], prob, logit)
    assert np.abs(logit.cpu().numpy() - O.forward(ocfg, params, ids, vals)["y"].numpy()).max() <= 1e-4
    eng.close()


@pytest.mark.parametrize("K,F,B", [(64, 13, 37), (64, 12, 520), (128, 9, 513), (256, 39, 37), (256, 6, 600), (72, 7, 530), (64, 6, 2100), (128, 20, 2050)])
def test_afm_wide_embeddings_pair_backward(K, F, B, dev):
    """K >= 64 (the reference runs AFM at K = 256, run.sh:18).  From 512 examples on the pair backward walks the pairs of an example
    in round-robin order (every row of d(pair tensor) read once; odd field counts have a bye), below that the two-reads kernel
    and the 1024-thread pooling kernels run; K = 72 pads to 128 physical columns; from 2048 examples on the forward pooling is the
    per-example U E product on the matrix cores (the backward pooling's G = (E diag(dy)) E^T runs at every size)."""
    V = 1500
    ocfg, params, eng = make_pair("afm", B=B, F=F, V=V, K=K, layers=(1,), att=(24,), opt="Adagrad", lr=1e-2)
    oopt = O.Optimizer(ocfg, params)
    for step in range(2):
        ids, vals, labels = O.synth_batch(B, F, V, seed=500 + step)
        ref_loss, _ = O.train_step(ocfg, params, oopt, ids, vals, labels)
        loss = eng.train_step(*dev_batch(ids, vals, labels, dev))
        assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (step, loss, ref_loss)
    got = eng.get_params()
    for name, ref in params.items():
        diff = np.abs(got[name] - ref.numpy()).max()
        assert diff <= 3e-6, (name, diff)
    eng.close()
    eng.close()


def test_device_feeder_wraps_the_input_slots(dev):
    """20 numpy batches through the 8 input slots (feeder thread + copy stream) train exactly like 20 direct steps."""
    from tf_repos_amd.feeder import DeviceFeeder
    F, V, B, K = 39, 3000, 128, 8
    ocfg, params, eng = make_pair("deepfm", B=B, F=F, V=V, K=K, layers=(32, 16), opt="Adagrad", lr=1e-2)
    _, _, ref = make_pair("deepfm", B=B, F=F, V=V, K=K, layers=(32, 16), opt="Adagrad", lr=1e-2)
    batches = [O.synth_batch(B if i != 19 else 50, F, V, seed=500 + i) for i in range(20)]
    for ids, vals, labels in batches:
        ref.train_step(*dev_batch(ids, vals, labels, dev), want_loss=False)
    fd = DeviceFeeder(eng, iter(batches))
    n = 0
    for ids, vals, labels, k in fd:
        assert ids.data_ptr() == eng.input_slot(k)[0].data_ptr()          # the step reads the slot in place
        eng.train_step(ids, vals, labels, want_loss=False)
        fd.release(k)
        n += 1
    fd.close()
    assert n == 20 and eng.global_step == 20
    a, b = eng.get_params(), ref.get_params()
    for name in a:
        assert np.abs(a[name] - b[name]).max() <= 1e-6, name
    eng.check_ids()
    eng.close(); ref.close()


def test_prefetched_id_grouping_changes_nothing(dev):
    """dctr_prefetch_ids (the input pipeline announcing the next batch) only moves the id grouping to the tail of the step in
    flight: losses and variables after four steps equal those of a run without the hint."""
    F, V, B = 39, 3000, 128
    runs = []
    for hint in (False, True):
        ocfg, params, eng = make_pair("deepfm", B=B, F=F, V=V, K=8, layers=(32, 16), opt="Adam", seed=3, use_graph=False)
        slots = []
        for i in range(4):
            ids, vals, labels = O.synth_batch(B, F, V, seed=700 + i)
            si, sv, sl = eng.input_slot(i)
            si[:B].copy_(torch.from_numpy(ids)); sv[:B].copy_(torch.from_numpy(vals)); sl[:B].copy_(torch.from_numpy(labels))
            slots.append((si[:B], sv[:B], sl[:B]))
        losses = []
        for i in range(4):
            losses.append(eng.train_step(*slots[i]))
            if hint and i + 1 < 4:
                eng.prefetch_ids(slots[i + 1][0])
        runs.append((losses, eng.get_params()))
        eng.close()
    assert np.allclose(runs[0][0], runs[1][0], rtol=0, atol=1e-6)
    for k in runs[0][1]:
        assert np.abs(runs[0][1][k] - runs[1][1][k]).max() <= 1e-6, k      # (the scatter's float atomics: two identical runs differ by ~1e-7 too)


@pytest.mark.parametrize("how", ["rewrite", "cancel", "staging_copy"])
def test_stale_prefetch_is_dropped(how, dev):
    """A hint whose batch is never trained (a training loop that stops on max_steps) must not leak into the next loop: the slot's
    generation (dctr_input_slot_rewrite, called by the feeder before a refill; the engine's own staging copy), or an explicit
    dctr_prefetch_cancel, invalidates it.  Without that the next step would scatter into the OLD batch's rows with no error."""
    F, V, B = 39, 3000, 128
    runs = []
    for hint in (False, True):
        ocfg, params, eng = make_pair("deepfm", B=B, F=F, V=V, K=8, layers=(32, 16), opt="Adam", seed=3, use_graph=False)
        b = [O.synth_batch(B, F, V, seed=800 + i) for i in range(3)]
        slot = 0 if how == "staging_copy" else 1                # (foreign buffers are staged into slot 0)
        s0, s1 = eng.input_slot(0), eng.input_slot(slot)

        def fill(s, batch):
            for dst, src in zip(s, batch):
                dst[:B].copy_(torch.from_numpy(src))
            torch.cuda.synchronize()
        if slot == 0:
            loss0 = eng.train_step(*dev_batch(*b[0], dev))
        else:
            fill(s0, b[0])
            loss0 = eng.train_step(s0[0][:B], s0[1][:B], s0[2][:B])
        fill(s1, b[1])
        if hint:
            eng.prefetch_ids(s1[0][:B])                         # announced ... and never trained
        if how == "rewrite":
            eng.input_slot_rewrite(slot)
            fill(s1, b[2])                                      # a new loop's feeder refills the slot: same address, same B
            loss1 = eng.train_step(s1[0][:B], s1[1][:B], s1[2][:B])
        elif how == "cancel":
            eng.prefetch_cancel()
            fill(s1, b[2])
            loss1 = eng.train_step(s1[0][:B], s1[1][:B], s1[2][:B])
        else:
            loss1 = eng.train_step(*dev_batch(*b[2], dev))      # staged by copy into slot 0, the slot that was announced
        runs.append((loss0, loss1, eng.get_params()))
        eng.close()
    assert abs(runs[0][0] - runs[1][0]) <= 1e-6 and abs(runs[0][1] - runs[1][1]) <= 1e-6
    for k in runs[0][2]:
        assert np.abs(runs[0][2][k] - runs[1][2][k]).max() <= 1e-6, k


@pytest.mark.parametrize("K", [128, 256])
def test_dcn_cross_wider_than_2560(K, dev):
    """--embedding_size 128 / 256 with 39 fields: F*K = 4992 / 9984 inputs to the cross network (DCN.py:140-145) -- one BLOCK per
    example instead of one wave (interact.hip)"""
    F, V, B = 39, 600, 24
    ocfg, params, eng = make_pair("dcn", B=B, F=F, V=V, K=K, layers=(32, 16), cross=3, opt="Adam", lr=1e-3, l2=1e-4, scale=0.02)
    ids, vals, labels = O.synth_batch(B, F, V, seed=41)
    d = dev_batch(ids, vals, labels, dev)
    ref = O.forward(ocfg, params, ids, vals)
    logit = torch.empty(B, device=dev)
    eng.predict(d[0], d[1], torch.empty(B, device=dev), logit)
    assert np.abs(logit.cpu().numpy() - ref["y"].numpy()).max() <= 1e-4
    oopt = O.Optimizer(ocfg, params)
    for step in range(2):
        ids, vals, labels = O.synth_batch(B, F, V, seed=42 + step)
        ref_loss, _ = O.train_step(ocfg, params, oopt, ids, vals, labels)
        loss = eng.train_step(*dev_batch(ids, vals, labels, dev))
        assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss))
    got = eng.get_params()
    for name, ref_p in params.items():
        assert np.abs(got[name] - ref_p.numpy()).max() <= 2e-6, name
    eng.close()


@pytest.mark.parametrize("K,cross", [(16, 3), (16, 1), (32, 4), (32, 2), (64, 3)])
def test_dcn_cross_step_pair(K, cross, dev):
    """The step's own cross-network kernels (interact.hip: a forward that keeps x_L and s_l only, a backward that re-forms the x_l and
    sums the cross parameters' gradients per block) at their three group sizes -- F*K = 624 (one wave per example), 1248 (two waves),
    2496 (a block) -- with 1..4 cross layers (DCN.py:150-158) and a batch that does not fill the last block: four Adam steps against
    the oracle, every variable, cross_w / cross_b included."""
    F, V, B = 39, 3000, 70
    ocfg, params, eng = make_pair("dcn", B=B, F=F, V=V, K=K, layers=(32, 16), cross=cross, opt="Adam", lr=1e-3, l2=1e-4, scale=0.02)
    oopt = O.Optimizer(ocfg, params)
    for step in range(4):
        ids, vals, labels = O.synth_batch(B if step != 2 else 37, F, V, seed=700 + step)
        ref_loss, _ = O.train_step(ocfg, params, oopt, ids, vals, labels)
        loss = eng.train_step(*dev_batch(ids, vals, labels, dev))
        assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (step, loss, ref_loss)
    got = eng.get_params()
    for name, ref_p in params.items():
        assert np.abs(got[name] - ref_p.numpy()).max() <= 3e-6, (name, float(np.abs(got[name] - ref_p.numpy()).max()))
    eng.close()


@pytest.mark.parametrize("model", MODELS)
@pytest.mark.parametrize("K", [10, 12, 24])
def test_any_embedding_size(model, K, dev):
    """--embedding_size is any integer in the reference (DeepFM.py:43); sizes the kernels do not take (K/4 not a power of two) run on
    the next one that they do, the extra columns held at zero, and every parameter is read and written in its logical shape."""
    F, V, B = (39, 2000, 64) if model != "afm" else (12, 800, 48)
    ocfg, params, eng = make_pair(model, B=B, F=F, V=V, K=K, layers=(32, 16), opt="Adam", lr=1e-2, l2=1e-3)
    assert eng.param_shapes["emb"] == (V, K)
    ids, vals, labels = O.synth_batch(B, F, V, seed=60)
    d = dev_batch(ids, vals, labels, dev)
    ref = O.forward(ocfg, params, ids, vals)
    logit = torch.empty(B, device=dev)
    eng.predict(d[0], d[1], torch.empty(B, device=dev), logit)
    assert np.abs(logit.cpu().numpy() - ref["y"].numpy()).max() <= 1e-4
    oopt = O.Optimizer(ocfg, params)
    for step in range(3):
        ids, vals, labels = O.synth_batch(B, F, V, seed=61 + step)
        ref_loss, _ = O.train_step(ocfg, params, oopt, ids, vals, labels)
        loss = eng.train_step(*dev_batch(ids, vals, labels, dev))
        assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (step, loss, ref_loss)
    got = eng.get_params()
    for name, ref_p in params.items():
        assert got[name].shape == tuple(ref_p.shape), name
        assert np.abs(got[name] - ref_p.numpy()).max() <= 3e-6, name     # (936-wide DCN inputs at K = 24: 2.2e-6 after three lr = 1e-2 Adam steps)
    assert np.abs(eng.get_slot("emb", 0) - oopt.slots["emb"]["m"].numpy()).max() <= 2e-6
    eng.close()


def test_unsupported_embedding_sizes_say_so(dev):
    from tf_repos_amd import errors
    from tf_repos_amd.engine import Engine, EngineConfig
    with pytest.raises(errors.UnimplementedError):
        Engine(EngineConfig(model="opnn", field_size=6, feature_size=100, embedding_size=10, deep_layers=(8,), dropout=(1.0,), max_batch=8))
    with pytest.raises(errors.InvalidArgumentError):
        Engine(EngineConfig(model="deepfm", field_size=6, feature_size=100, embedding_size=300, deep_layers=(8,), dropout=(1.0,), max_batch=8))


@pytest.mark.parametrize("B", [256, 4096])
def test_predict_and_stage_timing_right_behind_a_train_step(B, dev):
    """(round-5 ADVICE) The step's last join is deferred: the first layer's weight gradient, the MLP's optimizer launch and the weight re-split
    of step t may still be running on the side stream when the call returns.  A predict enqueued right behind it writes the gathered
    embeddings the deferred weight gradient reads (forward() joins first); dctr_time_kernel("train_step") CAPTURES steps (no deferred join
    inside a capture).  Both must leave the variables where an engine that synchronises after every call leaves them.
    B = 256: the lean small-batch step; B = 4096: the split-precision products with their planes."""
    F, V, K, layers = 39, 50_000, 16, (400, 400)
    kw = dict(model="deepfm", B=B, F=F, V=V, K=K, layers=layers, l2=1e-4, lr=5e-4, keep=(0.5, 0.5), use_graph=False, scale=0.01)
    batches = [dev_batch(*O.synth_batch(B, F, V, seed=70 + i), dev) for i in range(4)]
    states = []
    for sync in (True, False):
        _, _, eng = make_pair(**kw)
        prob = torch.empty(B, device=dev)
        for s in range(6):
            eng.train_step(*batches[s % 4], want_loss=False)
            if sync:
                torch.cuda.synchronize()
            eng.predict(batches[(s + 1) % 4][0], batches[(s + 1) % 4][1], prob, None)      # (writes x_in, reads every dense variable)
            if sync:
                torch.cuda.synchronize()
        states.append((dict(eng.get_params()), prob.cpu().numpy()))
        if not sync:
            assert eng.time_stage("train_step", iters=3) > 0.0          # captured steps: no unjoined stream at EndCapture
            eng.train_step(*batches[0], want_loss=True)                  # ... and the engine goes on
        eng.close()
    for k, v in states[0][0].items():
        assert np.abs(v - states[1][0][k]).max() <= 1e-6, k              # (the hot ids' float atomics: two runs differ by ~1e-8)
    assert np.abs(states[0][1] - states[1][1]).max() <= 1e-6
