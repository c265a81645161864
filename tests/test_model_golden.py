"""The model math pinned to the REFERENCE'S OWN SOURCE: tests/golden/models/*.npz were computed by evaluating, in numpy fp64,
the graph that the unmodified deep_ctr/Model_pipeline/*.py scripts build under the tf shim (tests/golden/make_model_golden.py,
oracle/graph_eval.py).  CPU: the torch restatement oracle/deepctr_oracle.py must reproduce them (fp64 to 1e-9, fp32 to 1e-5).
GPU: the HIP engine, through the C ABI, must reproduce them (logits <= 1e-4 abs as BASELINE.json's north_star states, measured
~1e-6; loss 1e-5 rel; every variable after two optimizer steps <= 5e-6 abs).
"""
import glob
import os
import sys

import numpy as np
import pytest
import torch

from oracle import deepctr_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from golden_util import draw_named, max_err, meta     # noqa: E402

CASES = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(HERE, "golden", "models", "*.npz")) if "serving" not in f)


def load(name):
    fx = dict(np.load(os.path.join(HERE, "golden", "models", name + ".npz"), allow_pickle=False))
    cfg = dict(meta(fx, "meta_config"))
    name_map = dict(meta(fx, "meta_name_map"))              # engine / oracle parameter name -> TF variable name
    shapes = dict(meta(fx, "meta_var_shapes"))
    var0 = draw_named(shapes, int(fx["meta_var_seed"]), float(fx["meta_var_scale"]))
    return fx, cfg, name_map, var0


def var_err(fx, s, t, got, gmin=0.0):
    """max |variable - expected| after step s; with gmin > 0 the elements whose expected gradient is smaller than gmin are left
    out: Adam divides by sqrt(v), so an fp32 rounding error on a 1e-7 gradient (a nearly dead ReLU under batch_norm) moves the
    variable by a visible fraction of lr -- in TF's fp32 as much as here."""
    key, gkey = "step%d/var/%s" % (s, t), "step%d/grad/%s" % (s, t)
    got = np.asarray(got, dtype=np.float64)
    if gmin <= 0.0 or not (gkey in fx or gkey + "@idx" in fx):
        return max_err(fx, key, got)
    if key in fx:
        keep = np.abs(fx[gkey]) >= gmin
        d = np.abs(got.reshape(fx[key].shape) - fx[key])
        return float(d[keep].max()) if keep.any() else 0.0
    keep = np.abs(fx[gkey + "@val"]) >= gmin
    d = np.abs(got.reshape(-1)[fx[key + "@idx"]] - fx[key + "@val"])
    return float(d[keep].max()) if keep.any() else 0.0


def masks_of(fx, s, as_torch=None):
    """{oracle dropout key: 0/1 mask} stored for step s (fixtures traced with keep_prob < 1), else None"""
    pre = "step%d/mask/" % s
    m = {k[len(pre):]: v for k, v in fx.items() if k.startswith(pre)}
    if not m:
        return None
    return {k: torch.from_numpy(v.astype(np.float64)).to(as_torch) for k, v in m.items()} if as_torch is not None else m


def oracle_config(cfg):
    keys = ("model", "field_size", "feature_size", "embedding_size", "deep_layers", "dropout", "attention_layers", "cross_layers",
            "l2_reg", "learning_rate", "optimizer", "batch_norm", "batch_norm_decay")
    return O.Config(**{k: cfg[k] for k in keys if k in cfg})


def test_fixtures_cover_every_deep_ctr_model():
    models = {load(c)[1]["model"] for c in CASES}
    assert {"deepfm", "fnn", "ipnn", "opnn", "nfm", "afm", "dcn", "mvm"} <= models
    opts = {str(load(c)[0]["meta_optimizer"]) for c in CASES}
    assert {"Adam", "Adagrad", "Momentum", "ftrl"} <= opts


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 2e-5)])
def test_oracle_reproduces_the_reference_graph(case, dtype, tol):
    fx, cfg, name_map, var0 = load(case)
    ocfg = oracle_config(cfg)
    shapes = O.param_shapes(ocfg)
    assert set(shapes) == set(name_map), (sorted(shapes), sorted(name_map))
    p = {e: torch.from_numpy(var0[t].reshape(shapes[e])).to(dtype) for e, t in name_map.items()}
    opt = O.Optimizer(ocfg, p)
    for s in range(int(fx["meta_steps"])):
        ids, vals, labels = fx["step%d/ids" % s], fx["step%d/vals" % s], fx["step%d/labels" % s]
        masks = masks_of(fx, s, dtype)
        out = O.forward(ocfg, p, ids, vals, train=True, masks=masks)
        assert np.abs(out["y"].double().numpy() - fx["step%d/logits" % s]).max() <= tol
        loss = float(O.loss_fn(ocfg, p, out["y"], torch.from_numpy(labels).to(dtype)))
        assert abs(loss - float(fx["step%d/loss" % s])) <= tol * max(1.0, abs(loss))
        g = O.grads(ocfg, p, ids, vals, labels, train=True, masks=masks)[1]
        for e, t in name_map.items():
            if "step%d/grad/%s" % (s, t) in fx or "step%d/grad/%s@idx" % (s, t) in fx:
                assert max_err(fx, "step%d/grad/%s" % (s, t), g[e].double().numpy()) <= tol, (s, t)
        O.train_step(ocfg, p, opt, ids, vals, labels, masks=masks)
        for e, t in name_map.items():
            assert var_err(fx, s, t, p[e].double().numpy(), gmin=(1e-4 if dtype == torch.float32 else 0.0)) <= tol * 5, (s, t)


def test_stored_dropout_masks_are_the_engine_function():
    """the masks in the keep_prob < 1 fixtures are dctr_dropout_mask(meta_engine_seed, step, site): what the HIP engine will draw"""
    from tf_repos_amd import capi
    sites = {"bi": capi.SITE_NFM_BI, "att": capi.SITE_AFM_ATT, "y_emb": capi.SITE_AFM_YEMB}
    sites.update({"mlp%d" % i: capi.SITE_MLP(i) for i in range(8)})
    seen = 0
    for case in CASES:
        fx, cfg, _, _ = load(case)
        for s in range(int(fx["meta_steps"])):
            for key, m in (masks_of(fx, s) or {}).items():
                keep = {"bi": cfg["dropout"][0], "att": cfg["dropout"][0], "y_emb": cfg["dropout"][1]}.get(key) or cfg["dropout"][int(key[3:])]
                again = np.empty(m.shape, np.uint8)
                capi.check(capi.lib().dctr_dropout_mask(int(fx["meta_engine_seed"]), s + 1, sites[key], m.size, float(keep), capi.ptr(again)))
                assert np.array_equal(again, m), (case, s, key)
                assert 0 < m.mean() < 1
                seen += 1
    assert seen >= 10


def test_oracle_on_the_reference_serving_sample():
    """deep_fm_serving_client.cpp:42-45, the only concrete example in the reference, through DeepFM.py in PREDICT mode at the
    README.md:49 operating point."""
    fx = dict(np.load(os.path.join(HERE, "golden", "models", "deepfm_serving_sample.npz"), allow_pickle=False))
    cfg, name_map = dict(meta(fx, "meta_config")), dict(meta(fx, "meta_name_map"))
    var0 = draw_named(dict(meta(fx, "meta_var_shapes")), int(fx["meta_var_seed"]), float(fx["meta_var_scale"]))
    ocfg = oracle_config(cfg)
    assert (ocfg.feature_size, ocfg.embedding_size, tuple(ocfg.deep_layers)) == (117581, 8, (400, 400, 400))
    shapes = O.param_shapes(ocfg)
    p = {e: torch.from_numpy(var0[t].reshape(shapes[e])).double() for e, t in name_map.items()}
    out = O.forward(ocfg, p, fx["ids"].reshape(1, 39), fx["vals"].reshape(1, 39), train=False)
    assert abs(float(out["y"][0]) - float(fx["logit"][0])) <= 1e-9 and abs(float(out["prob"][0]) - float(fx["prob"][0])) <= 1e-9


@pytest.mark.skipif(not os.path.isdir("/root/reference/deep_ctr"), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("case", ["deepfm_adam", "afm", "dcn", "afm_2att", "opnn_k16", "deepfm_dropout", "afm_dropout"])
def test_committed_fixtures_are_what_the_reference_source_produces_today(case, tmp_path):
    """Re-runs the generator on the reference tree (build container only) and requires the committed fixture, bit for bit."""
    import make_model_golden as gen
    i = [c[0] for c in gen.CASES].index(case)
    name, script, flags, params = gen.CASES[i]
    gen.run_case(name, script, flags, params, seed=i, out_dir=str(tmp_path), quiet=True)
    new = dict(np.load(os.path.join(str(tmp_path), name + ".npz")))
    old = dict(np.load(os.path.join(HERE, "golden", "models", name + ".npz")))
    assert set(new) == set(old)
    for k in old:
        if old[k].dtype.kind in "fiu":
            assert np.array_equal(new[k], old[k]), k


# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_engine_reproduces_the_reference_graph(case, dev):
    from tf_repos_amd.engine import Engine, EngineConfig
    fx, cfg, name_map, var0 = load(case)
    B = int(fx["step0/ids"].shape[0])
    dropout = "meta_engine_seed" in fx      # traced with keep_prob < 1: the engine draws the stored masks itself from this seed
    eng = Engine(EngineConfig(max_batch=B, seed=int(fx["meta_engine_seed"]) if dropout else 0, **cfg))
    for e, t in name_map.items():
        eng.set_param(e, var0[t].reshape(eng.param_shapes[e]))
    t_ = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for s in range(int(fx["meta_steps"])):
        ids, vals, labels = fx["step%d/ids" % s], fx["step%d/vals" % s], fx["step%d/labels" % s]
        # (a PREDICT pass of a batch-norm model uses the moving statistics, the TRAIN graph the batch's; and it has no dropout)
        if not cfg.get("batch_norm") and not dropout:
            prob, logit = torch.empty(B, device=dev), torch.empty(B, device=dev)
            eng.predict(t_(ids), t_(vals), prob, logit)
            assert np.abs(logit.cpu().numpy().astype(np.float64) - fx["step%d/logits" % s]).max() <= 1e-4
            assert np.abs(prob.cpu().numpy().astype(np.float64) - fx["step%d/prob" % s]).max() <= 1e-5
        loss = eng.train_step(t_(ids), t_(vals), t_(labels))
        assert abs(loss - float(fx["step%d/loss" % s])) <= 1e-5 * max(1.0, abs(loss)), (loss, float(fx["step%d/loss" % s]))
        for e, t in name_map.items():
            # (batch_norm, second step: the fp32 Adam step of the tiny-gradient elements excluded above feeds the next forward)
            assert var_err(fx, s, t, eng.get_param(e), gmin=1e-4) <= (2e-4 if cfg.get("batch_norm") and s > 0 else 5e-6), (s, t)
    eng.close()


@pytest.mark.gpu
def test_engine_on_the_reference_serving_sample(dev):
    from tf_repos_amd.engine import Engine, EngineConfig
    fx = dict(np.load(os.path.join(HERE, "golden", "models", "deepfm_serving_sample.npz"), allow_pickle=False))
    cfg, name_map = dict(meta(fx, "meta_config")), dict(meta(fx, "meta_name_map"))
    var0 = draw_named(dict(meta(fx, "meta_var_shapes")), int(fx["meta_var_seed"]), float(fx["meta_var_scale"]))
    eng = Engine(EngineConfig(max_batch=4, **cfg))
    for e, t in name_map.items():
        eng.set_param(e, var0[t].reshape(eng.param_shapes[e]))
    ids = torch.from_numpy(fx["ids"].reshape(1, 39).astype(np.int32)).to(dev)
    vals = torch.from_numpy(fx["vals"].reshape(1, 39)).to(dev)
    prob, logit = torch.empty(1, device=dev), torch.empty(1, device=dev)
    eng.predict(ids, vals, prob, logit)
    assert abs(float(logit[0]) - float(fx["logit"][0])) <= 1e-4 and abs(float(prob[0]) - float(fx["prob"][0])) <= 1e-5
    eng.close()
