"""Lowers the symbolic graph a reference-style model_fn builds onto the HIP engine.

The six deep_ctr model_fn's differ only in ~30 lines of interaction math, so lowering is a strict
structural recognizer: it identifies the tables, the interaction (FM second order / bi-interaction / inner or outer
product pairs / cross network / attention), the MLP stack, dropout keep_probs, the l2 terms of the loss and the
optimizer, and emits an EngineConfig plus the engine-name -> TF-variable-name map (checkpoint compatibility, SURVEY
Appendix A).  Anything it does not recognise raises -- there is no generic interpreter to fall back on.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

from .. import errors
from ..engine import EngineConfig
from . import graph as G


@dataclass
class Lowered:
    model: str
    config_kwargs: Dict
    name_map: Dict[str, str]                 # engine parameter name -> TF variable name
    predict_keys: List[str] = field(default_factory=lambda: ["prob"])
    # (DIN attention pooling: config_kwargs carries attention_layers + att_pairs)
    # CSR (multi-hot) models: the slot layout of the MLP input [(ids_key, vals_key | None, fixed_len)], the label keys
    # (["y"] / ["y", "z"]) and which engine output each predictions key / eval metric reads
    slots: Optional[List[Tuple[str, Optional[str], int]]] = None
    label_keys: List[str] = field(default_factory=list)
    outputs: Dict[str, int] = field(default_factory=dict)

    def engine_config(self, max_batch: int, **overrides) -> EngineConfig:
        kw = dict(self.config_kwargs)
        kw.update(overrides)
        return EngineConfig(max_batch=max_batch, **kw)


def _unsupported(msg):
    return errors.UnimplementedError("tf_repos_amd cannot lower this graph onto the HIP engine: " + msg)


def _through(t, ops=("reshape", "identity", "cast")):
    while isinstance(t, G.Tensor) and t.op in ops and t.inputs:
        t = t.inputs[0]
    return t


def lower(loss: Optional[G.Tensor], train_op: Optional[G.Tensor], predictions: Dict[str, G.Tensor], flags=None) -> Lowered:
    roots = [t for t in [loss, train_op] + list(predictions.values()) if isinstance(t, G.Tensor)]
    nodes = G.ancestors(roots)
    by_op: Dict[str, List[G.Tensor]] = {}
    for n in nodes:
        by_op.setdefault(n.op, []).append(n)
    bn_ops = by_op.get("batch_norm", [])
    if by_op.get("embedding_lookup_sparse") or by_op.get("iterator_varlen") or by_op.get("sparse_to_dense"):
        return _lower_multihot(loss, train_op, predictions, nodes, by_op)

    # ---- tables --------------------------------------------------------------------------------------------------
    lookups = by_op.get("embedding_lookup", [])
    emb_var = lin_var = None
    field_size = None
    for lk in lookups:
        var, ids = lk.inputs
        if not isinstance(var, G.Variable):
            raise _unsupported("embedding_lookup on a non-variable")
        if len(var.shape) == 2:
            emb_var = var
        elif len(var.shape) == 1:
            lin_var = var
        if ids.op == "reshape" and ids.attrs.get("shape") and len(ids.attrs["shape"]) == 2:
            field_size = int(ids.attrs["shape"][1])
    if emb_var is None or field_size is None:
        raise _unsupported("no [feature_size, embedding_size] embedding_lookup over reshape(feat_ids, [-1, field_size]) found")
    V, K = int(emb_var.shape[0]), int(emb_var.shape[1])
    F = field_size
    variables = {n.var_name: n for n in nodes if isinstance(n, G.Variable)}

    # ---- interaction -> model kind ---------------------------------------------------------------------------------
    fcs = by_op.get("fully_connected", [])
    hidden = [f for f in fcs if f.attrs["activation"] == "relu"]
    outs = [f for f in fcs if f.attrs["activation"] == "identity"]
    fc_vars = {id(v) for f in fcs for v in f.inputs[1:]}
    cross_vars = [v for v in variables.values() if len(v.shape) == 2 and v is not emb_var and id(v) not in fc_vars]
    bias_var = next((v for v in variables.values() if v.shape == (1,) and v.trainable and id(v) not in fc_vars), None)
    name_map: Dict[str, str] = {"emb": emb_var.var_name}
    kw: Dict = dict(field_size=F, feature_size=V, embedding_size=K)
    if by_op.get("softmax"):
        model = "afm"
    elif by_op.get("einsum"):
        model = "opnn"
    elif any(g.attrs.get("axis") == 1 and g.attrs.get("indices") for g in by_op.get("gather", [])):
        model = "ipnn"
    elif by_op.get("matmul") and len(cross_vars) == 2:
        model = "dcn"
    elif len(cross_vars) == 1 and tuple(cross_vars[0].shape) == (F, K) and lin_var is None:
        model = "mvm"           # DeepMVM.py:117-118,144-150: mvm_w [V,K] gathered, mvm_b [F,K] added, product over the fields
    elif not hidden:
        raise _unsupported("no hidden fully_connected layers")
    else:
        first_in = int(hidden[0].inputs[1].shape[0])
        has_sq = bool(by_op.get("square"))
        if has_sq and first_in == K:
            model = "nfm"
        elif has_sq and first_in == F * K:
            model = "deepfm"
        elif first_in == F * K:
            model = "fnn"
        else:
            raise _unsupported("first dense layer has fan-in %d (neither K=%d nor F*K=%d)" % (first_in, K, F * K))
    kw["model"] = model

    if model == "dcn":
        if lin_var is not None or bias_var is not None:
            raise _unsupported("DCN graph with a linear table / global bias")
        cw = next(v for v in cross_vars if any(n.op == "matmul" for n in nodes if _uses(n, v, nodes)))
        cb = next(v for v in cross_vars if v is not cw)
        if cw.shape != cb.shape or cw.shape[1] != F * K:
            raise _unsupported("cross_w/cross_b must both be [cross_layers, F*K]")
        kw["cross_layers"] = int(cw.shape[0])
        name_map["cross_w"], name_map["cross_b"] = cw.var_name, cb.var_name
    elif model == "mvm":
        if bias_var is not None:
            raise _unsupported("DeepMVM graph with a global bias")
        name_map["mvm_b"] = cross_vars[0].var_name
    else:
        if lin_var is None or bias_var is None:
            raise _unsupported("%s graph without linear table / global bias" % model)
        name_map["linear"], name_map["bias"] = lin_var.var_name, bias_var.var_name

    # ---- dense stack -----------------------------------------------------------------------------------------------
    if model == "afm":
        if bn_ops:
            raise _unsupported("batch_norm=True in the AFM graph is not implemented in the engine")
        att = [int(f.attrs["num_outputs"]) for f in hidden]
        kw["attention_layers"] = tuple(att)
        kw["deep_layers"] = (1,)
        for i, f in enumerate(hidden):
            name_map["att_mlp%d/weights" % i], name_map["att_mlp%d/biases" % i] = f.inputs[1].var_name, f.inputs[2].var_name
        if len(outs) != 2:
            raise _unsupported("AFM expects attention_out and deep_out projections")
        name_map["attention_out/weights"], name_map["attention_out/biases"] = outs[0].inputs[1].var_name, outs[0].inputs[2].var_name
        name_map["deep_out/weights"], name_map["deep_out/biases"] = outs[1].inputs[1].var_name, outs[1].inputs[2].var_name
    else:
        kw["deep_layers"] = tuple(int(f.attrs["num_outputs"]) for f in hidden)
        for i, f in enumerate(hidden):
            name_map["mlp%d/weights" % i], name_map["mlp%d/biases" % i] = f.inputs[1].var_name, f.inputs[2].var_name
        if bn_ops:
            # batch_norm_layer(x_deep, train_phase, scope_bn='bn_%d') after every hidden layer (DeepFM.py:159-160, 231-235)
            decays = set()
            for i, f in enumerate(hidden):
                mine = [b for b in bn_ops if _through(b.inputs[0]) is f]
                if len(mine) != 1:
                    raise _unsupported("expected one batch_norm on the output of hidden layer %d" % i)
                b = mine[0]
                if not (b.attrs["center"] and b.attrs["scale"]) or abs(b.attrs["epsilon"] - 1e-3) > 1e-12:
                    raise _unsupported("batch_norm must use center=True, scale=True, epsilon=0.001 (DeepFM.py:232-233)")
                decays.add(float(b.attrs["decay"]))
                for nm, v in zip(("beta", "gamma", "moving_mean", "moving_variance"), b.inputs[1:]):
                    name_map["bn_%d/%s" % (i, nm)] = v.var_name
            if len(decays) != 1:
                raise _unsupported("batch_norm layers with different decays")
            kw["batch_norm"], kw["batch_norm_decay"] = True, decays.pop()
        if len(outs) != 1 or int(outs[0].attrs["num_outputs"]) != 1:
            raise _unsupported("expected exactly one linear output layer of width 1")
        oname = "out_layer" if model == "dcn" else "deep_out"
        name_map[oname + "/weights"], name_map[oname + "/biases"] = outs[0].inputs[1].var_name, outs[0].inputs[2].var_name

    # ---- dropout (TRAIN graphs only; keep_prob semantics, DeepFM.py:162) -------------------------------------------------
    keep = [1.0] * len(kw["deep_layers"])
    pre_keep = None
    att_keep: List[float] = []
    for d in by_op.get("dropout", []):
        src = _through(d.inputs[0])
        if src.op == "batch_norm":
            src = _through(src.inputs[0])
        if src.op == "fully_connected" and src in hidden and model != "afm":
            keep[hidden.index(src)] = d.attrs["keep_prob"]
        elif model == "nfm":
            pre_keep = d.attrs["keep_prob"]
        elif model == "afm":
            att_keep.append(d.attrs["keep_prob"])
        else:
            raise _unsupported("dropout on %s" % src.op)
    if model == "nfm" and pre_keep is not None and abs(pre_keep - keep[0]) > 1e-12:
        raise _unsupported("NFM bi-interaction dropout (%g) must equal dropout[0] (%g) as in NFM.py:137,145" % (pre_keep, keep[0]))
    kw["dropout"] = tuple(att_keep) if model == "afm" else tuple(keep)

    # ---- loss: mean sigmoid_xent + l2_reg * l2_loss(table)  (DeepFM.py:188-190) ----------------------------------------------
    if loss is not None:
        if not by_op.get("sigmoid_xent"):
            raise _unsupported("loss is not sigmoid_cross_entropy_with_logits")
        regs = []
        for m in by_op.get("mul", []):
            a, b = m.inputs
            c, l = (a, b) if a.op == "const" else (b, a)
            if c.op == "const" and l.op == "l2_loss" and isinstance(l.inputs[0], G.Variable):
                regs.append((l.inputs[0].var_name, float(c.attrs["value"])))
        expect = {name_map[k] for k in {"dcn": ("cross_b", "cross_w", "emb"), "mvm": ("mvm_b", "emb")}.get(model, ("linear", "emb"))}
        if regs:
            coefs = {round(c, 12) for _, c in regs}
            if {n for n, _ in regs} != expect or len(coefs) != 1:
                raise _unsupported("l2_loss terms %s do not match the reference's set %s" % (sorted(regs), sorted(expect)))
            kw["l2_reg"] = regs[0][1]
        else:
            kw["l2_reg"] = 0.0

    # ---- optimizer (DeepFM.py:204-213) ---------------------------------------------------------------------------------------------
    if train_op is not None:
        if train_op.op != "minimize":
            raise _unsupported("train_op must come from optimizer.minimize(loss, global_step)")
        a = train_op.attrs
        h = a["hyper"]
        ok = {"Adam": h.get("beta1") == 0.9 and h.get("beta2") == 0.999 and h.get("epsilon") == 1e-8,
              "Adagrad": h.get("initial_accumulator_value") == 1e-8,
              "Momentum": h.get("momentum") == 0.95 and not h.get("use_nesterov"),
              "ftrl": h.get("learning_rate_power") == -0.5 and h.get("initial_accumulator_value") == 0.1 and not h.get("l1") and not h.get("l2")}
        if not ok.get(a["optimizer"], False):
            raise _unsupported("optimizer %s with hyper-parameters %s (engine implements the reference's settings only)" % (a["optimizer"], h))
        kw["optimizer"] = a["optimizer"]
        kw["learning_rate"] = a["learning_rate"]
    return Lowered(model=model, config_kwargs=kw, name_map=name_map, predict_keys=list(predictions.keys()))


def _optimizer_kwargs(train_op) -> Dict:
    if train_op.op != "minimize":
        raise _unsupported("train_op must come from optimizer.minimize(loss, global_step)")
    a = train_op.attrs
    h = a["hyper"]
    ok = {"Adam": h.get("beta1") == 0.9 and h.get("beta2") == 0.999 and h.get("epsilon") == 1e-8,
          "Adagrad": h.get("initial_accumulator_value") == 1e-8,
          "Momentum": h.get("momentum") == 0.95 and not h.get("use_nesterov"),
          "ftrl": h.get("learning_rate_power") == -0.5 and h.get("initial_accumulator_value") == 0.1 and not h.get("l1") and not h.get("l2")}
    if not ok.get(a["optimizer"], False):
        raise _unsupported("optimizer %s with hyper-parameters %s (engine implements the reference's settings only)" % (a["optimizer"], h))
    return {"optimizer": a["optimizer"], "learning_rate": a["learning_rate"]}


def _lower_multihot(loss, train_op, predictions, nodes, by_op) -> Lowered:
    """DIN with field-wise sum pooling (DIN.py:143-148,179-222) and ESMM (DeepCvrMTL.py:153-225): one shared embedding table,
    the MLP input = concat of [reshape(lookup(E, fixed ids [F'])) | lookup_sparse(E, ids, weights) ... | lookup(E, scalar id) ...],
    one tower + sigmoid xent (DIN) or a CTR and a CVR tower with the pCTCVR log-loss (ESMM)."""
    bn_ops = by_op.get("batch_norm", [])
    fcs = by_op.get("fully_connected", [])
    att_outs = [f for f in fcs if f.attrs["activation"] == "sigmoid"]            # att_out of the attention units (DIN.py:168)
    hidden = [f for f in fcs if f.attrs["activation"] == "relu"]
    outs = [f for f in fcs if f.attrs["activation"] == "identity"]
    # the MLP input is the concat that is not the [ub | ub - ax | ax] input of an attention unit
    unit_inputs = set()
    for ao in att_outs:
        unit_inputs.update(id(n) for n in G.ancestors([ao]) if n.op == "concat")
    concats = [c for c in by_op.get("concat", []) if c.attrs["axis"] == 1 and id(c) not in unit_inputs]
    if len(concats) != 1 or not hidden:
        raise _unsupported("expected one tf.concat(axis=1) feeding the MLP(s)")
    xcat = concats[0]
    units: List[Dict] = []           # attention units in concat order: {"part": index, "ad": tensor, "att_fcs": [...], "att_out": fc}

    # ---- slot layout, in concat order ----------------------------------------------------------------------------------------------
    emb_var = None
    slots: List[Tuple[str, Optional[str], int]] = []
    for part in xcat.inputs:
        src = _through(part, ops=("reshape", "identity"))
        if src.op == "embedding_lookup":
            var, ids = src.inputs
            if ids.op != "iterator_fixed":
                raise _unsupported("embedding_lookup ids must be a FixedLenFeature of the parsed Example")
            shp = ids.attrs["shape"]
            if len(shp) > 1:
                raise _unsupported("FixedLenFeature of rank %d" % len(shp))
            slots.append((ids.attrs["key"], None, int(shp[0]) if shp else 0))
        elif src.op == "reduce_sum" and att_outs:
            unit = _attention_unit(src, att_outs)
            var = unit["var"]
            unit["part"] = len(slots)
            units.append(unit)
            slots.append((unit["ids_key"], unit["vals_key"], -1))
        elif src.op == "embedding_lookup_sparse":
            var, ids = src.inputs[0], src.inputs[1]
            wts = src.inputs[2] if len(src.inputs) > 2 else None
            if ids.op != "iterator_varlen" or (wts is not None and wts.op != "iterator_varlen"):
                raise _unsupported("embedding_lookup_sparse ids / weights must be VarLenFeatures of the parsed Example")
            slots.append((ids.attrs["key"], wts.attrs["key"] if wts is not None else None, -1))
        else:
            raise _unsupported("MLP input part produced by %s" % src.op)
        if not isinstance(var, G.Variable) or (emb_var is not None and var is not emb_var):
            raise _unsupported("all lookups must read one shared embedding variable")
        emb_var = var
    V, K = int(emb_var.shape[0]), int(emb_var.shape[1])
    S = sum(n if n > 0 else 1 for _, _, n in slots)
    name_map: Dict[str, str] = {"emb": emb_var.var_name}
    att_kw: Dict = {}
    if units:
        # slot index of every concat part (a FixedLenFeature([n]) part spans n slots)
        first_slot, acc = [], 0
        for _, _, n in slots:
            first_slot.append(acc)
            acc += n if n > 0 else 1
        parts = [_through(p, ops=("reshape", "identity")) for p in xcat.inputs]
        pairs = []
        for u in units:
            ad_idx = [i for i, p in enumerate(parts) if p is _through(u["ad"], ops=("reshape", "identity"))]
            if len(ad_idx) != 1:
                raise _unsupported("an attention unit's ad embedding must also be one part of the MLP input (DIN.py:174-177,199)")
            pairs.append((first_slot[u["part"]], first_slot[ad_idx[0]]))
            if [id(v) for f in u["att_fcs"] + [u["att_out"]] for v in f.inputs[1:]] != \
               [id(v) for f in units[0]["att_fcs"] + [units[0]["att_out"]] for v in f.inputs[1:]]:
                raise _unsupported("the attention units must share their variables (variable_scope(reuse=tf.AUTO_REUSE), DIN.py:150)")
        u0 = units[0]
        for i, f in enumerate(u0["att_fcs"]):
            name_map["att_fc%d/weights" % i], name_map["att_fc%d/biases" % i] = f.inputs[1].var_name, f.inputs[2].var_name
        name_map["att_out/weights"], name_map["att_out/biases"] = u0["att_out"].inputs[1].var_name, u0["att_out"].inputs[2].var_name
        att_kw = dict(attention_layers=tuple(int(f.attrs["num_outputs"]) for f in u0["att_fcs"]), att_pairs=tuple(pairs))
        att_fc_ids = {id(f) for u in units for f in u["att_fcs"]}
        hidden = [f for f in hidden if id(f) not in att_fc_ids]

    # ---- towers ------------------------------------------------------------------------------------------------------------------------
    def chain(out_fc):
        """hidden layers from the concat to this output layer"""
        layers = []
        t = _through(out_fc.inputs[0], ops=("reshape", "identity", "dropout", "batch_norm"))
        while t is not xcat:
            if t.op != "fully_connected" or t.attrs["activation"] != "relu":
                raise _unsupported("tower contains %s" % t.op)
            layers.append(t)
            t = _through(t.inputs[0], ops=("reshape", "identity", "dropout", "batch_norm"))
        return layers[::-1]

    if any(int(o.attrs["num_outputs"]) != 1 for o in outs):
        raise _unsupported("output layers must have width 1")
    xents = by_op.get("sigmoid_xent", [])
    keep_of = {id(_through(d.inputs[0], ops=("reshape", "identity", "cast", "batch_norm"))): d.attrs["keep_prob"] for d in by_op.get("dropout", [])}
    kw: Dict = dict(field_size=S, feature_size=V, embedding_size=K)
    outputs: Dict[str, int] = {}
    if len(outs) == 1:
        model, label_keys = "din", ["y"]
        tower = chain(outs[0])
        towers = [("", "deep_out", tower, outs[0])]
        outputs = {k: 0 for k in predictions}            # {"prob": sigmoid(y)} (DIN.py:210-212)
    elif len(outs) == 2 and by_op.get("log_loss") or (len(outs) == 2 and loss is None):
        model, label_keys = "esmm", ["y", "z"]
        if loss is not None:
            if len(xents) != 1:
                raise _unsupported("ESMM loss must hold one sigmoid xent (CTR) and one log_loss (CTCVR)")
            ctr_out = _through(xents[0].inputs[0])
            if ctr_out not in outs:
                raise _unsupported("CTR logits are not an output layer")
        else:       # PREDICT graphs carry no loss: the towers are told apart by their variable scopes (DeepCvrMTL.py:195,203)
            ctr_out = next((o for o in outs if "ctr" in o.inputs[1].var_name), outs[1])
        cvr_out = next(o for o in outs if o is not ctr_out)
        towers = [("ctr_", "ctr_out", chain(ctr_out), ctr_out), ("cvr_", "cvr_out", chain(cvr_out), cvr_out)]
        if [int(f.attrs["num_outputs"]) for f in towers[0][2]] != [int(f.attrs["num_outputs"]) for f in towers[1][2]]:
            raise _unsupported("the CTR and CVR towers must have the same layer widths")
        # predictions (DeepCvrMTL.py:207-212): sigmoid of each tower's logit and their product
        for key, t in predictions.items():
            t0 = _through(t)
            if t0.op == "sigmoid":
                outputs[key] = 0 if _through(t0.inputs[0]) is ctr_out else 1
            elif t0.op == "mul":
                outputs[key] = 2
            else:
                raise _unsupported("prediction %r produced by %s" % (key, t0.op))
    else:
        raise _unsupported("%d output layers" % len(outs))
    keep = None
    bn_decays = set()
    for prefix, oname, layers, out in towers:
        for i, f in enumerate(layers):
            name_map["%smlp%d/weights" % (prefix, i)], name_map["%smlp%d/biases" % (prefix, i)] = f.inputs[1].var_name, f.inputs[2].var_name
        name_map[oname + "/weights"], name_map[oname + "/biases"] = out.inputs[1].var_name, out.inputs[2].var_name
        k2 = tuple(float(keep_of.get(id(f), 1.0)) for f in layers)
        if keep is not None and k2 != keep:
            raise _unsupported("the towers use different dropout keep_probs")
        keep = k2
        if bn_ops:
            # batch_norm_layer(x, train_phase, scope_bn) after every hidden ReLU of the tower (DIN.py:203-204, DeepCvrMTL.py:177-178)
            if units:
                raise _unsupported("batch_norm together with attention pooling (DIN.py:165 uses an undefined train_phase there)")
            for i, f in enumerate(layers):
                mine = [b for b in bn_ops if _through(b.inputs[0]) is f]
                if len(mine) != 1:
                    raise _unsupported("expected one batch_norm on the output of %smlp%d" % (prefix, i))
                b = mine[0]
                if not (b.attrs["center"] and b.attrs["scale"]) or abs(b.attrs["epsilon"] - 1e-3) > 1e-12:
                    raise _unsupported("batch_norm must use center=True, scale=True, epsilon=0.001")
                bn_decays.add(float(b.attrs["decay"]))
                for nm, v in zip(("beta", "gamma", "moving_mean", "moving_variance"), b.inputs[1:]):
                    name_map["%sbn_%d/%s" % (prefix, i, nm)] = v.var_name
    if bn_ops:
        if len(bn_decays) != 1:
            raise _unsupported("batch_norm layers with different decays")
        kw["batch_norm"], kw["batch_norm_decay"] = True, bn_decays.pop()
    kw.update(model=model, deep_layers=tuple(int(f.attrs["num_outputs"]) for f in towers[0][2]), dropout=keep)
    if units:
        if model != "din":
            raise _unsupported("attention pooling outside the DIN graph")
        for u in units:         # the units' dropout reuses dropout[i] of the deep layers (DIN.py:166-167)
            for i, f in enumerate(u["att_fcs"]):
                if abs(float(keep_of.get(id(f), 1.0)) - keep[i]) > 1e-12:
                    raise _unsupported("attention layer %d keep_prob differs from dropout[%d]" % (i, i))
        kw.update(att_kw)

    # ---- loss --------------------------------------------------------------------------------------------------------------------------
    if loss is not None:
        consts = {}
        for m in by_op.get("mul", []):
            a, b = m.inputs
            c, l = (a, b) if a.op == "const" else (b, a)
            if c.op != "const":
                continue
            l0 = _through(l, ops=("reshape", "identity", "reduce_mean"))
            if l0.op == "l2_loss" and l0.inputs[0] is emb_var:
                consts["l2"] = float(c.attrs["value"])
            elif l0.op == "sigmoid_xent":
                consts["ctr"] = float(c.attrs["value"])
            elif l0.op == "log_loss":
                consts["cvr"] = float(c.attrs["value"])
        kw["l2_reg"] = consts.get("l2", 0.0)
        if model == "esmm":
            if "ctr" not in consts or "cvr" not in consts or abs(consts["ctr"] + consts["cvr"] - 1.0) > 1e-6:
                raise _unsupported("ESMM loss must be w * ctr_loss + (1 - w) * cvr_loss (DeepCvrMTL.py:225)")
            ll = by_op["log_loss"][0]
            if abs(ll.attrs["epsilon"] - 1e-7) > 1e-15:
                raise _unsupported("log_loss epsilon must be the default 1e-7")
            kw["ctr_task_wgt"] = consts["ctr"]
        elif len(xents) != 1:
            raise _unsupported("DIN loss is not one sigmoid_cross_entropy_with_logits")
        # which parsed label feeds which task: xent labels = clicks (y), log_loss labels = conversions (z)
        def label_key(t):
            t = _through(t)
            if t.op != "iterator_fixed":
                raise _unsupported("labels must be FixedLenFeature([], float32) outputs of the parsed Example")
            return t.attrs["key"]
        label_keys = [label_key(xents[0].inputs[1])]
        if model == "esmm":
            label_keys.append(label_key(by_op["log_loss"][0].inputs[0]))
    if train_op is not None:
        kw.update(_optimizer_kwargs(train_op))
    return Lowered(model=model, config_kwargs=kw, name_map=name_map, predict_keys=list(predictions.keys()), slots=slots,
                   label_keys=label_keys, outputs=outputs)


def _attention_unit(pooled: G.Tensor, att_outs) -> Dict:
    """One attention_unit (DIN.py:152-172) from its output reduce_sum(dense_emb * att_wgt * dense_mask, 1):
    which parsed features it reads, the ad embedding it is queried with, its layers."""
    if pooled.attrs.get("axis") != 1:
        raise _unsupported("attention unit must pool over axis 1")
    anc = G.ancestors([pooled])
    ops: Dict[str, List[G.Tensor]] = {}
    for n in anc:
        ops.setdefault(n.op, []).append(n)
    dense = ops.get("sparse_to_dense", [])
    ids = [d for d in dense if d.dtype in (G.int64, G.int32)]
    vals = [d for d in dense if d.dtype is G.float32]
    lookups = [l for l in ops.get("embedding_lookup", []) if l.inputs[1] in ids]
    if len(ids) != 1 or len(vals) != 1 or len(lookups) != 1:
        raise _unsupported("attention unit must read one id list and one weight list through sparse_tensor_to_dense")
    outs = [a for a in att_outs if a in anc]
    tiles = ops.get("tile", [])
    if len(outs) != 1 or len(tiles) != 1:
        raise _unsupported("attention unit must hold one sigmoid output layer and one tiled ad embedding")
    cat = [c for c in ops.get("concat", []) if len(c.inputs) == 3]
    if len(cat) != 1:
        raise _unsupported("attention unit input must be concat([ub, ub - ax, ax])")
    a, d, b = cat[0].inputs
    if not (d.op == "sub" and d.inputs[0] is a and d.inputs[1] is b and _through(b) is tiles[0] and _through(a).op == "mul"):
        raise _unsupported("attention unit input is not [ub | ub - ax | ax] (DIN.py:162)")
    gr = ops.get("greater", [])
    if len(gr) != 1 or gr[0].inputs[0] is not ids[0] or gr[0].inputs[1].op != "const" or float(gr[0].inputs[1].attrs["value"]) != 0.0:
        raise _unsupported("attention unit mask must be dense_ids > 0 (DIN.py:157)")
    layers = []
    t = _through(outs[0].inputs[0], ops=("reshape", "identity", "dropout"))
    while t is not cat[0]:
        if t.op != "fully_connected" or t.attrs["activation"] != "relu":
            raise _unsupported("attention MLP contains %s" % t.op)
        layers.append(t)
        t = _through(t.inputs[0], ops=("reshape", "identity", "dropout"))
    if int(outs[0].attrs["num_outputs"]) != 1:
        raise _unsupported("att_out must have width 1")
    return {"var": lookups[0].inputs[0], "ids_key": ids[0].inputs[0].attrs["key"], "vals_key": vals[0].inputs[0].attrs["key"],
            "ad": tiles[0].inputs[0], "att_fcs": layers[::-1], "att_out": outs[0]}


def _uses(node, var, nodes) -> bool:
    """True if `node` consumes `var` directly or through getitem/reshape."""
    for i in node.inputs:
        j = i
        while isinstance(j, G.Tensor) and j.op in ("getitem", "reshape"):
            j = j.inputs[0]
        if j is var:
            return True
    return False
