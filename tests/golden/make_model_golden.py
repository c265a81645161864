#!/usr/bin/env python
"""Generates tests/golden/models/*.npz: expected logits / loss / gradients / post-step variables computed FROM THE
REFERENCE'S OWN model_fn SOURCE.  Run in the build container (needs /root/reference; it does not exist on the GPU box):

    python tests/golden/make_model_golden.py

For every case the unmodified script (deep_ctr/Model_pipeline/{DeepFM,PNN,NFM,AFM,DCN,DeepMVM}.py; the only edits are the
three in-memory py2->py3 fixes of tf_repos_amd/run_reference.py) is executed under the tf shim, which records its tf.* calls
as a symbolic graph; oracle/graph_eval.py evaluates that graph in numpy fp64 (values + reverse-mode gradients) on seeded
inputs and weights, and applies the optimizer the script asked for.  The per-op meaning of the TF calls is assumed (SURVEY
Appendix B); everything else -- which tensors meet, pair order, axes, what is regularised -- is the reference's.

The fixtures pin oracle/deepctr_oracle.py (tests/test_model_golden.py, CPU) and the HIP engine (same file, -m gpu).
"""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/deep_ctr"
OUT = os.path.join(ROOT, "tests", "golden", "models")

from oracle.graph_eval import GraphEval, optimizer_step     # noqa: E402


def synth(B, F, V, seed):
    """Criteo-shaped batch: 13 numeric fields (ids 1..13, values in [0,1)), the rest categorical with repeated hot ids."""
    rng = np.random.default_rng(seed)
    ids = np.zeros((B, F), np.int32)
    vals = np.ones((B, F), np.float32)
    nnum = min(13, F // 3)
    for f in range(F):
        if f < nnum:
            ids[:, f] = min(f + 1, V - 1)
            vals[:, f] = np.round(rng.random(B), 6).astype(np.float32)
        else:
            lo = nnum + 1 + (f - nnum) * max(1, (V - nnum - 1) // (F - nnum))
            span = max(1, (V - nnum - 1) // (F - nnum))
            z = np.minimum((rng.zipf(1.3, B) - 1), span - 1)
            ids[:, f] = np.minimum(lo + z, V - 1)
    labels = (rng.random(B) < 0.3).astype(np.float32)
    return ids, vals, labels


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from golden_util import draw_named, store     # noqa: E402


def draw_variables(variables, seed, scale):
    return draw_named({k: tuple(v.shape) for k, v in variables.items() if k != "global_step"}, seed, scale)


def trace(script, flags, params, mode="train", batch=8):
    import tf_repos_amd.tf_shim as shim
    from tf_repos_amd.run_reference import load_reference_module
    from tf_repos_amd.tf_shim import graph as G
    from tf_repos_amd.tf_shim.lowering import lower
    mod = load_reference_module(os.path.join(REF, "Model_pipeline", script))
    for k, v in flags.items():
        setattr(shim.FLAGS_MODULE.FLAGS, k, v)
    tf = sys.modules["tensorflow"]
    est = tf.estimator.Estimator(model_fn=mod.model_fn, model_dir="/tmp/unused", params=params)
    g = G.Graph()
    with g:
        feats, labels = mod.input_fn(["/tmp/none.libsvm"], num_epochs=1, batch_size=batch)
        spec = est._call_model_fn(feats, labels if mode != "infer" else None, mode)
        roots = [t for t in [spec.loss, spec.train_op] + list((spec.predictions or {}).values()) if t is not None]
        nodes = G.ancestors(roots)
        lowered = lower(spec.loss, spec.train_op, spec.predictions or {}, shim.FLAGS_MODULE.FLAGS)
        variables = dict(g.variables)
    return spec, nodes, feats, labels, variables, lowered


def _fc_scope(node):
    """the fully_connected scope ('mlp0', 'cvr_mlp1', 'att_fc0', ...) whose output (optionally through batch_norm) `node` is, or None"""
    while node is not None and getattr(node, "op", None) in ("batch_norm",):
        node = node.inputs[0]
    if getattr(node, "op", None) != "fully_connected":
        return None
    return node.inputs[1].var_name.split("/")[-2]


def engine_mask(seed, step, site, shape, keep):
    """the 0/1 mask the HIP engine applies at `site` in train step `step` (host function of the C ABI; no GPU involved)"""
    from tf_repos_amd import capi
    m = np.empty(shape, np.uint8)
    capi.check(capi.lib().dctr_dropout_mask(int(seed), int(step), int(site), int(m.size), float(keep), capi.ptr(m)))
    return m


class FixedFieldMasks:
    """nn.dropout sites of the fixed-field scripts in the engine's numbering (include/deepctr_hip.h "dropout sites"): the layer
    outputs (DeepFM.py:161-162) by the scope of the fully_connected they follow, NFM's bi-interaction (NFM.py:136-137), AFM's
    attention weights and pooled embedding in call order (AFM.py:152-153,157-158).  Called by GraphEval for every dropout node."""

    def __init__(self, model, seed, step):
        self.model, self.seed, self.step, self.used, self.n_other = model, seed, step, {}, 0

    def __call__(self, node, shape):
        from tf_repos_amd import capi
        scope = _fc_scope(node.inputs[0])
        if scope is not None and re.fullmatch(r"mlp\d+", scope):
            key, site = scope, capi.SITE_MLP(int(scope[3:]))
        elif self.model == "nfm":
            key, site = "bi", capi.SITE_NFM_BI
        elif self.model == "afm":
            key, site = [("att", capi.SITE_AFM_ATT), ("y_emb", capi.SITE_AFM_YEMB)][self.n_other]
            self.n_other += 1
        else:
            raise ValueError("unexpected dropout node %s in model %s" % (node.name, self.model))
        assert key not in self.used, key
        self.used[key] = engine_mask(self.seed, self.step, site, shape, node.attrs["keep_prob"])
        return self.used[key]


def run_case(name, script, flags, params, B=24, steps=2, seed=0, var_scale=0.05, out_dir=None, quiet=False):
    spec, nodes, feats, labels, variables, lowered = trace(script, flags, params)
    with_dropout = any(n.op == "dropout" and n.attrs["keep_prob"] < 1.0 for n in nodes)
    F, V = int(params["field_size"]), int(params["feature_size"])
    var0 = draw_variables(variables, 1000 + seed, var_scale)
    mini = [n for n in nodes if n.op == "minimize"][0]
    kind, lr, hyper = mini.attrs["optimizer"], mini.attrs["learning_rate"], mini.attrs["hyper"]
    prob = spec.predictions["prob"]
    assert prob.op == "sigmoid"
    logit = prob.inputs[0]
    out = {"meta_script": script, "meta_flags": repr(sorted(flags.items())), "meta_model": lowered.model,
           "meta_optimizer": kind, "meta_lr": lr, "meta_hyper": repr(sorted(hyper.items())),
           "meta_config": repr(sorted((k, v) for k, v in lowered.config_kwargs.items())),
           "meta_name_map": repr(sorted(lowered.name_map.items())), "meta_steps": steps, "meta_var_seed": 1000 + seed,
           "meta_var_scale": var_scale, "meta_var_shapes": repr(sorted((k, tuple(v.shape)) for k, v in var0.items()))}
    var = {k: v.astype(np.float64) for k, v in var0.items()}
    slots = {}
    for s in range(steps):
        ids, vals, lab = synth(B, F, V, seed=7000 + 10 * seed + s)
        masks = FixedFieldMasks(lowered.model, 1000 + seed, s + 1) if with_dropout else None
        ev = GraphEval(nodes, var, training=True, dropout_masks=masks)
        ev.eval({feats["feat_ids"]: ids.reshape(B, F, 1), feats["feat_vals"]: vals.reshape(B, F, 1), labels: lab})
        g = ev.grad(spec.loss)
        if with_dropout:        # the engine draws these itself from (meta_engine_seed, step, site); the oracle is handed them
            out["meta_engine_seed"] = 1000 + seed
            for k, m in masks.used.items():
                out["step%d/mask/%s" % (s, k)] = m
        out["step%d/ids" % s], out["step%d/vals" % s], out["step%d/labels" % s] = ids, vals, lab
        out["step%d/logits" % s] = ev.val[logit.id].reshape(-1)
        out["step%d/prob" % s] = ev.val[prob.id].reshape(-1)
        out["step%d/loss" % s] = np.float64(ev.val[spec.loss.id])
        for k, gv in g.items():
            store(out, "step%d/grad/%s" % (s, k), gv)
        new = optimizer_step(kind, lr, hyper, var, g, slots, t=s + 1)
        var.update(new)
        var.update(ev.bn_updates)
        for k, v in var.items():
            store(out, "step%d/var/%s" % (s, k), v)
    os.makedirs(out_dir or OUT, exist_ok=True)
    np.savez_compressed(os.path.join(out_dir or OUT, name + ".npz"), **out)
    if not quiet:
        print("%-22s %-12s model=%-6s opt=%-8s loss0=%.9f  |logit|max=%.4f  vars=%d" % (name, script, lowered.model, kind, out["step0/loss"],
              np.abs(out["step0/logits"]).max(), len(var0)))


def serving_sample():
    """The one concrete 39-field example of the reference: Serving_pipeline/deep_fm_serving_client.cpp:42-45 (ids < 117581)."""
    src = open(os.path.join(REF, "Serving_pipeline", "deep_fm_serving_client.cpp")).read()
    ids = re.search(r"ids_vec\s*=\s*\{([^}]*)\}", src, re.S).group(1)
    vals = re.search(r"vals_vec\s*=\s*\{([^}]*)\}", src, re.S).group(1)
    ids = np.array([int(x) for x in ids.replace("\n", " ").split(",")], np.int32)
    vals = np.array([float(x) for x in vals.replace("\n", " ").split(",")], np.float32)
    assert ids.shape == (39,) and vals.shape == (39,)
    return ids, vals


def run_serving_case():
    """PREDICT-mode DeepFM at the README.md:49 operating point (feature_size 117581, 400-400-400) on the serving sample.  The
    weights are NOT stored (3.8 MB): both sides draw them with draw_variables(seed 4242)."""
    params = dict(field_size=39, feature_size=117581, embedding_size=8, learning_rate=0.0005, batch_norm_decay=0.9, l2_reg=1e-4,
                  deep_layers="400,400,400", dropout="0.5,0.5,0.5")
    spec, nodes, feats, labels, variables, lowered = trace("DeepFM.py", {}, params, mode="infer", batch=1)
    var = draw_variables(variables, 4242, 0.05)
    ids, vals = serving_sample()
    ev = GraphEval(nodes, var, training=False)
    ev.eval({feats["feat_ids"]: ids.reshape(1, 39, 1), feats["feat_vals"]: vals.reshape(1, 39, 1)})
    prob = spec.predictions["prob"]
    out = {"ids": ids, "vals": vals, "logit": ev.val[prob.inputs[0].id].reshape(-1), "prob": ev.val[prob.id].reshape(-1),
           "meta_var_shapes": repr(sorted((k, tuple(v.shape)) for k, v in var.items())), "meta_name_map": repr(sorted(lowered.name_map.items())),
           "meta_var_seed": 4242, "meta_var_scale": 0.05,
           "meta_config": repr(sorted((k, v) for k, v in lowered.config_kwargs.items()))}
    np.savez_compressed(os.path.join(OUT, "deepfm_serving_sample.npz"), **out)
    print("deepfm_serving_sample  logit=%.9f prob=%.9f" % (out["logit"][0], out["prob"][0]))


BASE = dict(field_size=39, feature_size=400, embedding_size=4, learning_rate=0.01, batch_norm_decay=0.9, l2_reg=1e-3,
            deep_layers="16,8", dropout="1.0,1.0,1.0", cross_layers=2, attention_layers="6")

CASES = [
    ("deepfm_adam", "DeepFM.py", {}, BASE),
    ("deepfm_adagrad", "DeepFM.py", {"optimizer": "Adagrad"}, BASE),
    ("deepfm_momentum", "DeepFM.py", {"optimizer": "Momentum"}, BASE),
    ("deepfm_ftrl", "DeepFM.py", {"optimizer": "ftrl"}, BASE),
    ("deepfm_bn", "DeepFM.py", {"batch_norm": True}, BASE),
    ("fnn", "PNN.py", {"model_type": "FNN"}, BASE),
    ("ipnn", "PNN.py", {"model_type": "Inner"}, BASE),
    ("opnn", "PNN.py", {"model_type": "Outer"}, BASE),
    ("nfm", "NFM.py", {}, dict(BASE, dropout="1.0,1.0,1.0")),
    ("afm", "AFM.py", {}, dict(BASE, dropout="1.0,1.0")),
    ("dcn", "DCN.py", {}, BASE),
    ("mvm", "DeepMVM.py", {}, BASE),
    # the operating point of deep_ctr/README.md:49 (K = 8 as BASELINE c1), vocabulary shrunk so the fixture stays small
    ("deepfm_c1_shape", "DeepFM.py", {}, dict(BASE, feature_size=2000, embedding_size=8, learning_rate=0.0005, l2_reg=1e-4, deep_layers="400,400,400")),
    # (appended, so that the seeds of the cases above stay what they were)
    # AFM.py:143-145 with two attention widths; Outer-PNN at K = 16, where the engine forms the pair products inside the GEMMs
    ("afm_2att", "AFM.py", {}, dict(BASE, dropout="1.0,1.0", attention_layers="12,6")),
    ("opnn_k16", "PNN.py", {"model_type": "Outer"}, dict(BASE, field_size=8, feature_size=300, embedding_size=16)),
    # TRAIN graphs WITH dropout (keep_prob < 1: every operating point of run.sh / README.md:49): the masks are the ones the HIP
    # engine draws (a pure function of seed, step, site and element index, evaluated on the host), stored in the fixture
    ("deepfm_dropout", "DeepFM.py", {}, dict(BASE, dropout="0.5,0.5,0.5")),
    ("nfm_dropout", "NFM.py", {}, dict(BASE, dropout="0.5,0.8,0.8")),
    ("afm_dropout", "AFM.py", {}, dict(BASE, dropout="0.7,0.6")),
    ("deepfm_bn_dropout", "DeepFM.py", {"batch_norm": True, "optimizer": "Momentum"}, dict(BASE, dropout="0.8,0.5,0.5")),
    ("dcn_dropout", "DCN.py", {}, dict(BASE, dropout="0.8,0.8")),
]

if __name__ == "__main__":
    if not os.path.isdir(REF):
        raise SystemExit("needs the reference tree at " + REF)
    only = sys.argv[1:]
    for i, (name, script, flags, params) in enumerate(CASES):
        if only and name not in only:
            continue
        run_case(name, script, flags, params, seed=i, B=(256 if name == "deepfm_c1_shape" else 24))
    if not only or "deepfm_serving_sample" in only:
        run_serving_case()
