"""The multi-hot models pinned to the REFERENCE'S OWN SOURCE: tests/golden/models_csr/*.npz were computed by evaluating, in numpy
fp64, the graph that the unmodified deep_ctr/Model_pipeline/DIN.py (sum pooling and attention pooling, DIN.py:143-222) and
DeepMTL/Model_pipeline/DeepCvrMTL.py (ESMM, DeepCvrMTL.py:153-225) build under the tf shim
(tests/golden/make_multihot_golden.py, oracle/graph_eval.py).  CPU: oracle/multihot_oracle.py must reproduce them (fp64 to
1e-9, fp32 to 2e-5).  GPU: the HIP engine's CSR path, through the C ABI, must reproduce them (outputs 1e-4 / 1e-5, loss 1e-5 rel,
every variable after two optimizer steps <= 5e-6 abs).  Cases with keep_prob < 1 carry the engine's own dropout masks."""
import glob
import os
import sys

import numpy as np
import pytest
import torch

from oracle import multihot_oracle as M

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from golden_util import draw_named, max_err, meta     # noqa: E402
from tests.test_model_golden import var_err           # noqa: E402

DIR = os.path.join(HERE, "golden", "models_csr")
CASES = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(DIR, "*.npz")))


def load(name):
    fx = dict(np.load(os.path.join(DIR, name + ".npz"), allow_pickle=False))
    cfg = dict(meta(fx, "meta_config"))
    name_map = dict(meta(fx, "meta_name_map"))
    var0 = draw_named(dict(meta(fx, "meta_var_shapes")), int(fx["meta_var_seed"]), float(fx["meta_var_scale"]))
    return fx, cfg, name_map, var0


def oracle_config(fx, cfg):
    return M.Config(model=cfg["model"], field_size=int(fx["meta_common_fields"]), feature_size=cfg["feature_size"],
                    embedding_size=cfg["embedding_size"], deep_layers=tuple(cfg["deep_layers"]), dropout=tuple(cfg["dropout"]),
                    l2_reg=cfg["l2_reg"], learning_rate=cfg["learning_rate"], optimizer=cfg["optimizer"],
                    ctr_task_wgt=cfg.get("ctr_task_wgt", 0.5), attention_layers=tuple(cfg["attention_layers"]) if cfg.get("att_pairs") else (),
                    batch_norm=bool(cfg.get("batch_norm", False)), batch_norm_decay=cfg.get("batch_norm_decay", 0.9))


def batch_of(fx, s):
    p = "step%d/" % s
    b = {"feat_ids": fx[p + "feat_ids"], "y": fx[p + "y"], "z": fx[p + "z"]}
    for n in M.MULTI_W:
        b[n] = (fx[p + n + "/off"], fx[p + n + "/ids"], fx[p + n + "/vals"])
    for n in M.SINGLE:
        b[n] = fx[p + n]
    for n in M.MULTI_NW:
        b[n] = (fx[p + n + "/off"], fx[p + n + "/ids"], None)
    return b


def masks_of(fx, s, dtype):
    """oracle keys: '<tower>mlp<i>' [B, H]; '<unit>/att_fc<i>' over the unit's real entries (the fixture also holds the padded form
    the script's graph consumed)"""
    pre = "step%d/mask/" % s
    m = {}
    for k, v in fx.items():
        if k.startswith(pre):
            k = k[len(pre):]
            if "/att_fc" in k and not k.endswith("@entries"):
                continue
            m[k.replace("@entries", "")] = torch.from_numpy(v.astype(np.float64)).to(dtype)
    return m or None


def test_fixtures_cover_din_and_esmm():
    cfgs = [load(c)[1] for c in CASES]
    assert {"din", "esmm"} <= {c["model"] for c in cfgs}
    assert any(c.get("att_pairs") for c in cfgs) and any(c.get("batch_norm") for c in cfgs) and any(min(c["dropout"]) < 1 for c in cfgs)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-9), (torch.float32, 2e-5)])
def test_multihot_oracle_reproduces_the_reference_graph(case, dtype, tol):
    fx, cfg, name_map, var0 = load(case)
    ocfg = oracle_config(fx, cfg)
    shapes = M.param_shapes(ocfg)
    assert set(shapes) == set(name_map), (sorted(shapes), sorted(name_map))
    p = {e: torch.from_numpy(var0[t].reshape(shapes[e])).to(dtype) for e, t in name_map.items()}
    opt = M.Optimizer(ocfg, p)
    for s in range(int(fx["meta_steps"])):
        batch = batch_of(fx, s)
        masks = masks_of(fx, s, dtype)
        out = M.forward(ocfg, p, batch, train=True, masks=masks)
        for k in meta(fx, "meta_outputs"):
            assert np.abs(out[k].double().numpy() - fx["step%d/out/%s" % (s, k)]).max() <= tol, (s, k)
        if ocfg.model == "din":
            assert np.abs(out["y"].double().numpy() - fx["step%d/out/logit" % s]).max() <= tol
        loss, g, _ = M.grads(ocfg, p, batch, train=True, masks=masks)
        assert abs(float(loss) - float(fx["step%d/loss" % s])) <= tol * max(1.0, abs(float(loss)))
        for e, t in name_map.items():
            if "step%d/grad/%s" % (s, t) in fx:
                assert max_err(fx, "step%d/grad/%s" % (s, t), g[e].double().numpy()) <= tol, (s, t)
        M.train_step(ocfg, p, opt, batch, masks=masks)
        for e, t in name_map.items():
            assert var_err(fx, s, t, p[e].double().numpy(), gmin=(1e-4 if dtype == torch.float32 else 0.0)) <= tol * 5, (s, t)


def test_stored_masks_are_the_engine_function():
    from tf_repos_amd import capi
    seen = 0
    for case in CASES:
        fx, cfg, _, _ = load(case)
        if "meta_engine_seed" not in fx:
            continue
        Fc, B = int(fx["meta_common_fields"]), fx["step0/feat_ids"].shape[0]
        S = Fc + 8
        for s in range(int(fx["meta_steps"])):
            b = batch_of(fx, s)
            off, _, _ = M.slot_csr(oracle_config(fx, cfg), b)
            slot_of = np.repeat(np.arange(B * S) % S, np.diff(off))
            for k, m in fx.items():
                if not k.startswith("step%d/mask/" % s) or ("/att_fc" in k and not k.endswith("@entries")):
                    continue
                key = k.split("/mask/")[1]
                if key.endswith("@entries"):                # '<unit>/att_fc<i>@entries': rows of the engine's [nnz, A] mask of the unit's slot
                    unit, layer = key[:-len("@entries")].split("/att_fc")
                    i = int(layer)
                    full = np.empty((int(off[-1]), m.shape[1]), np.uint8)
                    capi.check(capi.lib().dctr_dropout_mask(int(fx["meta_engine_seed"]), s + 1, capi.SITE_MLP2(i), full.size,
                                                            float(cfg["dropout"][i]), capi.ptr(full)))
                    again = full[slot_of == Fc + M.MULTI_W.index(unit)]
                else:
                    i = int(key[-1])
                    site = capi.SITE_MLP2(i) if key.startswith("cvr_") else capi.SITE_MLP(i)
                    again = np.empty(m.shape, np.uint8)
                    capi.check(capi.lib().dctr_dropout_mask(int(fx["meta_engine_seed"]), s + 1, site, m.size, float(cfg["dropout"][i]), capi.ptr(again)))
                assert np.array_equal(again, m), (case, k)
                seen += 1
    assert seen >= 12


@pytest.mark.skipif(not os.path.isfile("/root/reference/deep_ctr/Model_pipeline/DIN.py"), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("case", ["din_sum", "din_att", "esmm", "din_att_dropout"])
def test_committed_fixtures_are_what_the_reference_source_produces_today(case, tmp_path):
    import make_multihot_golden as gen
    i = [c[0] for c in gen.CASES].index(case)
    name, path, flags, params = gen.CASES[i]
    gen.run_case(name, path, dict(gen.RESET, **flags), params, seed=i, out_dir=str(tmp_path), quiet=True)
    new = dict(np.load(os.path.join(str(tmp_path), name + ".npz")))
    old = dict(np.load(os.path.join(DIR, name + ".npz")))
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
    ocfg = oracle_config(fx, cfg)
    B = int(fx["step0/feat_ids"].shape[0])
    dropout = "meta_engine_seed" in fx
    eng = Engine(EngineConfig(max_batch=B, max_entries=B * (ocfg.n_slots + 40), seed=int(fx["meta_engine_seed"]) if dropout else 0, **cfg))
    for e, t in name_map.items():
        eng.set_param(e, var0[t].reshape(eng.param_shapes[e]))
    t_ = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    esmm = cfg["model"] == "esmm"
    bn = bool(cfg.get("batch_norm"))
    for s in range(int(fx["meta_steps"])):
        batch = batch_of(fx, s)
        off, ids, wts = M.slot_csr(ocfg, batch)
        d_off, d_ids, d_wts = t_(off), t_(ids), t_(wts)
        if not bn and not dropout:          # PREDICT = the TRAIN graph's forward only without batch statistics and dropout
            o = [torch.empty(B, device=dev) for _ in range(3)]
            eng.predict_csr(d_off, d_ids, d_wts, B, *o)
            torch.cuda.synchronize()
            got = dict(zip(("pctr", "pcvr", "pctcvr"), o)) if esmm else {"prob": o[0], "logit": o[1]}
            for k, v in got.items():
                assert np.abs(v.cpu().numpy().astype(np.float64) - fx["step%d/out/%s" % (s, k)]).max() <= (1e-4 if k == "logit" else 1e-5), (s, k)
        loss = eng.train_step_csr(d_off, d_ids, d_wts, t_(batch["y"]), t_(batch["z"]) if esmm else None)
        assert abs(loss - float(fx["step%d/loss" % s])) <= 1e-5 * max(1.0, abs(loss)), (s, loss, float(fx["step%d/loss" % s]))
        for e, t in name_map.items():
            assert var_err(fx, s, t, eng.get_param(e), gmin=1e-4) <= (2e-4 if bn and s > 0 else 5e-6), (s, t)
    eng.close()
