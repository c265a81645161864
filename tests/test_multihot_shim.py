"""The TF-1.x surface of the multi-hot models.  CPU: the reference's DIN.py (--attention_pooling=False) and DeepCvrMTL.py
(read from /root/reference when present) and examples/multihot_estimator.py trace and lower onto the din / esmm engine models
with the slot layout of their tf.concat.  GPU: Estimator.train / evaluate / predict over TFRecord files match the oracle."""
import importlib.util
import os
import sys

import numpy as np
import pytest

from oracle import deepctr_oracle as O
from oracle import multihot_oracle as M
from tests.test_tfrecord import write_file
from tf_repos_amd import errors

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIN = "/root/reference/deep_ctr/Model_pipeline/DIN.py"
REF_ESMM = "/root/reference/DeepMTL/Model_pipeline/DeepCvrMTL.py"
PARAMS = dict(field_size=11, feature_size=5000, embedding_size=8, learning_rate=0.0005, batch_norm_decay=0.9, l2_reg=1e-4,
              deep_layers="64,32", dropout="0.5,0.5", attention_layers="256", ctr_task_wgt=0.4)
SLOTS = [("feat_ids", None, 11), ("u_catids", "u_catvals", -1), ("u_shopids", "u_shopvals", -1), ("u_brandids", "u_brandvals", -1),
         ("u_intids", "u_intvals", -1), ("a_catids", None, 0), ("a_shopids", None, 0), ("a_brandids", None, 0), ("a_intids", None, -1)]


def _trace(model_fn, input_fn, params, mode="train"):
    tf = sys.modules["tensorflow"]
    est = tf.estimator.Estimator(model_fn=model_fn, model_dir="/tmp/unused", params=params)
    return est, est._build(input_fn, mode)


@pytest.mark.skipif(not os.path.isfile(REF_DIN), reason="reference tree not present (GPU box)")
def test_reference_din_and_esmm_scripts_lower_onto_the_engine():
    from tf_repos_amd.run_reference import load_reference_module
    import tf_repos_amd.tf_shim as shim
    mod = load_reference_module(REF_DIN)
    shim.FLAGS_MODULE.FLAGS.field_size = 11
    fn = lambda: mod.input_fn(["/tmp/none.tfrecord"], num_epochs=1, batch_size=256)
    # the default --attention_pooling=True (DIN.py:45): four attention units sharing att_fc0 / att_out; the script sizes att_fc%d
    # with the DEEP widths layers[i] (DIN.py:164), here 64, not with --attention_layers=256
    est, (spec, low, pipe, variables) = _trace(mod.model_fn, fn, PARAMS)
    assert low.model == "din" and low.slots == SLOTS
    assert low.config_kwargs["attention_layers"] == (64,) and low.config_kwargs["att_pairs"] == ((11, 15), (12, 16), (13, 17), (14, 18))
    assert low.name_map["att_fc0/weights"] == "Field-wise-Pooling-layer/att_fc0/weights"
    assert low.name_map["att_out/biases"] == "Field-wise-Pooling-layer/att_out/biases"
    shim.FLAGS_MODULE.FLAGS.attention_pooling = False
    est, (spec, low, pipe, variables) = _trace(mod.model_fn, fn, PARAMS)
    assert low.model == "din" and low.slots == SLOTS and low.label_keys == ["y"] and "att_pairs" not in low.config_kwargs
    kw = low.config_kwargs
    assert (kw["field_size"], kw["feature_size"], kw["embedding_size"]) == (19, 5000, 8)          # 11 + 8 slots
    assert kw["deep_layers"] == (64, 32) and kw["dropout"] == (0.5, 0.5) and kw["l2_reg"] == pytest.approx(1e-4)
    assert low.name_map["emb"] == "embeddings" and low.name_map["deep_out/weights"] == "DIN-out/din_out/weights"
    assert low.name_map["mlp1/biases"] == "MLP-layer/mlp1/biases"
    assert pipe.tfrecord and [(s.ids_feature, s.vals_feature, s.fixed_len) for s in pipe.slot_specs] == SLOTS

    shim.FLAGS_MODULE.FLAGS.batch_norm = True              # --batch_norm=True: bn_%d after every hidden ReLU (DIN.py:203-204)
    est, (spec, low, pipe, variables) = _trace(mod.model_fn, fn, PARAMS)
    assert low.config_kwargs["batch_norm"] is True and low.config_kwargs["batch_norm_decay"] == 0.9
    assert low.name_map["bn_1/moving_variance"] == "MLP-layer/bn_1/moving_variance" and low.config_kwargs["dropout"] == (0.5, 0.5)

    mod = load_reference_module(REF_ESMM)
    shim.FLAGS_MODULE.FLAGS.field_size = 11
    fn = lambda: mod.input_fn(["/tmp/none.tfrecord"], num_epochs=1, batch_size=256)
    est, (spec, low, pipe, variables) = _trace(mod.model_fn, fn, PARAMS)
    assert low.model == "esmm" and low.slots == SLOTS and low.label_keys == ["y", "z"]
    assert low.config_kwargs["ctr_task_wgt"] == pytest.approx(0.4) and low.outputs == {"pcvr": 1, "pctr": 0, "pctcvr": 2}
    # tf.name_scope("CVR_Task") does not prefix variable names (DeepCvrMTL.py:166,174)
    assert low.name_map["cvr_mlp0/weights"] == "cvr_mlp0/weights" and low.name_map["ctr_out/biases"] == "ctr_out/biases"
    est, (spec, low, pipe, variables) = _trace(mod.model_fn, fn, PARAMS, "eval")
    assert est._metric_outputs(spec, low) == {"CTR_AUC": 0, "CVR_AUC": 1, "CTCVR_AUC": 2}
    shim.FLAGS_MODULE.FLAGS.batch_norm = True              # scopes cvr_bn_%d / ctr_bn_%d (DeepCvrMTL.py:178,199)
    est, (spec, low, pipe, variables) = _trace(mod.model_fn, fn, PARAMS)
    assert low.config_kwargs["batch_norm"] is True and low.name_map["cvr_bn_0/gamma"] == "cvr_bn_0/gamma"
    assert low.name_map["ctr_bn_1/moving_mean"] == "ctr_bn_1/moving_mean"
    shim.FLAGS_MODULE.FLAGS.batch_norm = False
    with pytest.raises(TypeError):           # PREDICT passes labels=None and the script indexes labels['y'] (DeepCvrMTL.py:146): as in TF
        _trace(mod.model_fn, fn, PARAMS, "infer")


def _load_example():
    import tf_repos_amd.tf_shim as shim
    shim.install()
    shim.FLAGS_MODULE.FLAGS._reset()
    spec = importlib.util.spec_from_file_location("multihot_estimator_example", os.path.join(ROOT, "examples", "multihot_estimator.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("task", ["din", "din_att", "esmm"])
def test_example_script_lowers(task):
    mod = _load_example()
    p = dict(PARAMS, optimizer="Momentum", attention_layers="24,12" if task == "din_att" else "")
    att = task == "din_att"
    task = "din" if att else task
    est = mod.build_estimator(task, p, "/tmp/unused")
    fn = lambda: mod.input_fn(["/tmp/none"], 64, 1, field_size=11, with_z=task == "esmm")
    for mode in ("train", "eval", "infer"):
        spec, low, pipe, variables = est._build(fn, mode)
        assert low.model == task and low.slots == SLOTS
        assert low.config_kwargs.get("attention_layers") == ((24, 12) if att else None)
    spec, low, pipe, variables = est._build(fn, "train")
    if att:
        assert low.config_kwargs["att_pairs"] == ((11, 15), (12, 16), (13, 17), (14, 18))
        assert low.name_map["att_fc1/weights"] == "pooling/score_fc1/weights" and low.name_map["att_out/weights"] == "pooling/score_out/weights"
    assert low.config_kwargs["optimizer"] == "Momentum" and low.config_kwargs["dropout"] == (0.5, 0.5)


@pytest.mark.gpu
@pytest.mark.parametrize("task", ["din", "din_att", "esmm"])
def test_estimator_over_tfrecords_matches_oracle(task, tmp_path, dev):
    import torch
    mod = _load_example()
    Fc, V, K, B = 5, 900, 8, 32
    att = (12,) if task == "din_att" else ()
    task = "din" if att else task
    ocfg = M.Config(model=task, field_size=Fc, feature_size=V, embedding_size=K, deep_layers=(32, 16), dropout=(1.0, 1.0), l2_reg=1e-3,
                    learning_rate=0.01, optimizer="Adam", ctr_task_wgt=0.3, attention_layers=att)
    train = [M.synth_batch(ocfg, n, seed=60 + i) for i, n in enumerate((B, B, B, 11))]          # ragged last batch
    valid = [M.synth_batch(ocfg, B, seed=70 + i) for i in range(2)]
    for i, b in enumerate(train):
        write_file(tmp_path / ("tr%d.tfrecord" % i), b)
    for i, b in enumerate(valid):
        write_file(tmp_path / ("va%d.tfrecord" % i), b)
    p = dict(field_size=Fc, feature_size=V, embedding_size=K, learning_rate=0.01, l2_reg=1e-3, deep_layers="32,16", dropout="1.0,1.0",
             optimizer="Adam", ctr_task_wgt=0.3, attention_layers=",".join(str(a) for a in att))
    est = mod.build_estimator(task, p, str(tmp_path / "ckpt"), log_steps=2)
    files = lambda pre, n: [str(tmp_path / ("%s%d.tfrecord" % (pre, i))) for i in range(n)]
    tr_fn = lambda: mod.input_fn(files("tr", 4), B, 1, field_size=Fc, with_z=task == "esmm")
    va_fn = lambda: mod.input_fn(files("va", 2), B, 1, field_size=Fc, with_z=task == "esmm")
    spec, low, pipe, variables = est._build(tr_fn, "train")
    est._ensure_engine(low, variables, B, est._csr_capacity(pipe))
    names = {"din": {"deep_out": "din_out", "mlp": "din_mlp"}}.get(task, {})
    params = {k: torch.from_numpy(est._engine.get_param(k).copy()) for k in M.param_shapes(ocfg)}
    assert set(low.name_map) == set(params)
    est.train(input_fn=tr_fn)
    opt = M.Optimizer(ocfg, params)
    for b in train:
        M.train_step(ocfg, params, opt, b)
    for ename, tfname in low.name_map.items():
        assert np.abs(est.get_variable_value(tfname) - params[ename].numpy()).max() <= 3e-6, tfname
    assert est._engine.global_step == 4
    # EVAL
    res = est.evaluate(input_fn=va_fn)
    outs = [M.forward(ocfg, params, b) for b in valid]
    tot = sum(float(M.loss_fn(ocfg, params, o, b)) - ocfg.l2_reg * float(M.l2_loss(params["emb"])) for o, b in zip(outs, valid)) / len(valid)
    assert abs(res["loss"] - (tot + ocfg.l2_reg * float(M.l2_loss(params["emb"])))) <= 1e-5
    pairs = {"din": [("auc", "y", "prob")], "esmm": [("CTR_AUC", "y", "pctr"), ("CVR_AUC", "z", "pcvr"), ("CTCVR_AUC", "z", "pctcvr")]}[task]
    for key, lab, pred in pairs:
        auc = O.StreamingAUC()
        for o, b in zip(outs, valid):
            auc.update(b[lab], o[pred].numpy())
        assert abs(res[key] - auc.result()) <= 1e-5, key
    # PREDICT
    preds = list(est.predict(input_fn=va_fn))
    assert len(preds) == 2 * B
    for key in pairs:
        got = np.array([d[key[2]] for d in preds])
        want = np.concatenate([o[key[2]].numpy() for o in outs])
        assert np.abs(got - want).max() <= 1e-5, key


REF_TFR = "/root/reference/DeepMTL/Feature_pipeline/get_tfrecord.py"


@pytest.mark.skipif(not os.path.isfile(REF_TFR), reason="reference tree not present (GPU box)")
def test_reference_tfrecord_writer_feeds_the_c_parser(tmp_path):
    """Feature_pipeline/get_tfrecord.py (tf.python_io.TFRecordWriter + tf.train.Example, run unchanged under the shim) writes the
    joined Ali-CCP sample lines as TFRecords; the C parser reads them back in the slot layout of DIN.py."""
    from tf_repos_amd.run_reference import load_reference_module
    from tf_repos_amd import tfrecord as T
    import tf_repos_amd.tf_shim as shim
    mod = load_reference_module(REF_TFR)
    # two sample lines in the format of get_tfrecord.py:43: sample_id,y,z,"field:id:value ..."
    lines = ["40362692,0,0,216:9342395:1.0 301:9351665:1.0 205:7702673:1.0 206:8317829:1.0 207:8967741:1.0 508:9356012:2.30259 "
             "210:9059239:1.0 210:9042796:1.0 210:9076972:1.0 127_14:3529789:2.3979 127_14:3806412:2.70805",
             "40362693,1,1,101:11:1.0 121:12:1.0 109_14:500:0.5 109_14:501:1.5 110_14:600:2.0 150_14:700:1.0 206:801:1.0 216:901:1.0"]
    src = tmp_path / "part-00000"
    src.write_text("\n".join(lines) + "\nmalformed,line\n")
    shim.FLAGS_MODULE.FLAGS.output_dir = str(tmp_path)
    mod.gen_tfrecords(str(src))
    buf = (tmp_path / "part-00000.tfrecord").read_bytes()
    specs = [T.SlotSpec("feat_ids", None, 11), T.SlotSpec("u_catids", "u_catvals"), T.SlotSpec("u_shopids", "u_shopvals"),
             T.SlotSpec("u_brandids", "u_brandvals"), T.SlotSpec("u_intids", "u_intvals"), T.SlotSpec("a_catids", None, 0),
             T.SlotSpec("a_shopids", None, 0), T.SlotSpec("a_brandids", None, 0), T.SlotSpec("a_intids", None, -1)]
    off, ids, wts, labels = T.parse_slot_csr(buf, specs, ["y", "z"], 10_000_000)
    assert labels.tolist() == [[0.0, 1.0], [0.0, 1.0]] and len(off) == 2 * 19 + 1
    S = 19
    slot = lambda b, s: (ids[off[b * S + s]:off[b * S + s + 1]].tolist(), wts[off[b * S + s]:off[b * S + s + 1]].tolist())
    # example 0: common fields 205 / 301 present, the others take their default ids 1..11 (get_tfrecord.py:34,64-70)
    common0 = [slot(0, s)[0][0] for s in range(11)]
    assert sorted(common0) == sorted([1, 2, 3, 4, 5, 6, 7, 8, 9, 7702673, 9351665])
    assert slot(0, 13) == ([3529789, 3806412], [np.float32(2.3979), np.float32(2.70805)])        # u_brand = field 127_14 with its values
    assert slot(0, 11) == ([12], [1.0])                                                            # u_cat absent: default id 12, weight 1
    assert slot(0, 15)[0] == [8317829] and slot(0, 16)[0] == [8967741] and slot(0, 17)[0] == [9342395]   # a_cat 206, a_shop 207, a_brand 216
    assert slot(0, 18)[0] == [9059239, 9042796, 9076972]                                           # a_int = field 210, multi-hot
    # example 1
    assert slot(1, 11) == ([500, 501], [0.5, 1.5]) and slot(1, 12) == ([600], [2.0]) and slot(1, 14) == ([700], [1.0])
    assert slot(1, 15)[0] == [801] and slot(1, 16)[0] == [17] and slot(1, 17)[0] == [901] and slot(1, 18)[0] == [18]
