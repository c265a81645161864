"""tf.feature_column + the canned estimators wide_n_deep.py uses (wide_n_deep.py:92-151), served by the engine's
wide / deep / wide_n_deep models (DCTR_MODEL_WIDE / DEEP / WND).

    tf.feature_column.numeric_column / categorical_column_with_identity / embedding_column / make_parse_example_spec
    tf.estimator.LinearClassifier / DNNClassifier / DNNLinearCombinedClassifier
    tf.estimator.export.build_parsing_serving_input_receiver_fn

What the TF-1.4 canned estimators do with these columns is restated in oracle/canned_oracle.py's header (every rule
tagged [TF-1.4]; the reference pins none of it).  Mapping onto the engine:
  * the categorical identity columns share ONE stacked table (column c owns rows [offset_c, offset_c + num_buckets_c));
    ids outside [0, num_buckets) take default_value before the offset is added;
  * the numeric columns are the engine's dense inputs;
  * TF concatenates the DNN input columns in NAME-SORTED order; the engine keeps [embeddings | numeric] -- the only
    observable difference is the row order of dnn/hiddenlayer_0/kernel, permuted here when variables are read or set by
    their TF names.
"""
from __future__ import annotations

import glob
import json
import math
import os
import time
from typing import Any, Dict, Iterator, List, Optional, Sequence

import numpy as np

from .. import errors
from ..engine import Engine, EngineConfig
from . import data as D
from . import graph as G
from .estimator import RunConfig


# ---- feature columns -------------------------------------------------------------------------------------------------
class NumericColumn:
    def __init__(self, key, shape=(1,), default_value=None, dtype=None, normalizer_fn=None):
        if tuple(shape) != (1,) or normalizer_fn is not None:
            raise errors.UnimplementedError("numeric_column: only scalar columns without normalizer_fn")
        self.key = self.name = key


class IdentityCategoricalColumn:
    def __init__(self, key, num_buckets, default_value=None):
        self.key = self.name = key
        self.num_buckets = int(num_buckets)
        self.default_value = default_value


class EmbeddingColumn:
    def __init__(self, categorical_column, dimension, combiner="mean", initializer=None, **_kw):
        if not isinstance(categorical_column, IdentityCategoricalColumn):
            raise errors.UnimplementedError("embedding_column over %r" % (categorical_column,))
        self.categorical_column = categorical_column
        self.dimension = int(dimension)
        self.combiner = combiner
        self.key = categorical_column.key
        self.name = categorical_column.key + "_embedding"


def numeric_column(key, **kw):
    return NumericColumn(key, **kw)


def categorical_column_with_identity(key, num_buckets, default_value=None):
    return IdentityCategoricalColumn(key, num_buckets, default_value)


def embedding_column(categorical_column, dimension, **kw):
    return EmbeddingColumn(categorical_column, dimension, **kw)


def make_parse_example_spec(feature_columns) -> Dict[str, Dict[str, Any]]:
    """FixedLenFeature([1], float32) for numeric columns, VarLenFeature(int64) for categorical ones [TF-1.4]."""
    spec: Dict[str, Dict[str, Any]] = {}
    for c in feature_columns:
        if isinstance(c, NumericColumn):
            spec[c.key] = {"kind": "FixedLenFeature", "shape": [1], "dtype": "float32"}
        elif isinstance(c, (IdentityCategoricalColumn, EmbeddingColumn)):
            spec[c.key] = {"kind": "VarLenFeature", "dtype": "int64"}
        else:
            raise errors.UnimplementedError("feature column %r" % (c,))
    return spec


class _ParsingReceiver:
    def __init__(self, feature_spec):
        self.feature_spec = feature_spec


def build_parsing_serving_input_receiver_fn(feature_spec, default_batch_size=None):
    return lambda: _ParsingReceiver(feature_spec)


# ---- estimators ------------------------------------------------------------------------------------------------------
def _truncated_normal(rng, shape, std):
    a = rng.normal(0, std, size=shape)
    bad = np.abs(a) > 2 * std
    while bad.any():
        a[bad] = rng.normal(0, std, size=int(bad.sum()))
        bad = np.abs(a) > 2 * std
    return a.astype(np.float32)


class _CannedEstimator:
    """Shared body of LinearClassifier / DNNClassifier / DNNLinearCombinedClassifier (binary head)."""

    def __init__(self, model_type: str, model_dir, linear_columns, dnn_columns, hidden_units, config, n_classes=2):
        if n_classes != 2:
            raise errors.UnimplementedError("only the binary head (n_classes=2) is implemented")
        self.model_type = model_type
        self._config = config or RunConfig()
        self.model_dir = model_dir or self._config.model_dir or "/tmp/tf_repos_amd_model"
        self.hidden_units = [int(h) for h in (hidden_units or [])]
        cols = list(linear_columns or []) + list(dnn_columns or [])
        self.numeric: List[NumericColumn] = []
        self.categorical: List[IdentityCategoricalColumn] = []
        self.dimension = 4
        for c in cols:
            if isinstance(c, NumericColumn):
                if c.key not in [n.key for n in self.numeric]:
                    self.numeric.append(c)
            elif isinstance(c, (IdentityCategoricalColumn, EmbeddingColumn)):
                cat = c.categorical_column if isinstance(c, EmbeddingColumn) else c
                if isinstance(c, EmbeddingColumn):
                    self.dimension = c.dimension
                if cat.key not in [k.key for k in self.categorical]:
                    self.categorical.append(cat)
            else:
                raise errors.UnimplementedError("feature column %r" % (c,))
        dims = {c.dimension for c in cols if isinstance(c, EmbeddingColumn)}
        if len(dims) > 1:
            raise errors.UnimplementedError("embedding columns of different dimensions")
        self.offsets = np.concatenate([[0], np.cumsum([c.num_buckets for c in self.categorical])]).astype(np.int64)
        self.n_linear_columns = len(list(linear_columns or []))
        self._engine: Optional[Engine] = None

    # -- TF-1.4 defaults (oracle/canned_oracle.py header) -------------------------------------------------------------------
    @property
    def dnn_learning_rate(self) -> float:
        return 0.05 if self.model_type == "deep" else 0.001

    @property
    def linear_learning_rate(self) -> float:
        cap = 0.2 if self.model_type == "wide" else 0.005
        return min(cap, 1.0 / math.sqrt(max(1, self.n_linear_columns)))

    @property
    def config(self):
        return self._config

    # -- engine ---------------------------------------------------------------------------------------------------------------
    def _ensure_engine(self, batch_size: int) -> Engine:
        if self._engine is not None and self._engine.cfg.max_batch >= batch_size:
            return self._engine
        state = self._snapshot() if self._engine is not None else self._load_latest()
        if self._engine is not None:
            self._engine.close()
        n = len(self.hidden_units)
        self._engine = Engine(EngineConfig(
            model=self.model_type, field_size=len(self.categorical), feature_size=int(self.offsets[-1]),
            embedding_size=self.dimension, deep_layers=tuple(self.hidden_units), dropout=(1.0,) * n, l2_reg=0.0,
            learning_rate=self.dnn_learning_rate, optimizer="Adagrad", table_mode="touched_rows", max_batch=batch_size,
            seed=int(self._config.tf_random_seed or 0), dense_size=len(self.numeric), lin_optimizer="ftrl",
            lin_learning_rate=self.linear_learning_rate, loss_sum=True))
        if state is not None:
            self._restore(state)
        else:
            self._initialize()
        return self._engine

    def _initialize(self) -> None:
        """linear_model: zeros; embeddings: truncated normal(0, 1/sqrt(dim)); dense kernels glorot-uniform, biases 0 [TF-1.4]."""
        e = self._engine
        rng = np.random.default_rng(int(self._config.tf_random_seed or 0))
        for name, shp in e.param_shapes.items():
            if name == "emb":
                v = _truncated_normal(rng, shp, 1.0 / math.sqrt(self.dimension))
            elif name.endswith("/weights"):
                lim = math.sqrt(6.0 / (shp[0] + shp[1]))
                v = rng.uniform(-lim, lim, size=shp).astype(np.float32)
            else:
                v = np.zeros(shp, dtype=np.float32)
            e.set_param(name, v)

    # -- variables under their TF-1.4 names ----------------------------------------------------------------------------------
    def _dnn_input_order(self) -> np.ndarray:
        """TF's input_layer concatenates columns sorted by name; returns, for each row of TF's first kernel, the engine's row."""
        K = self.dimension
        blocks = []
        for i, c in enumerate(self.categorical):
            blocks.append((c.key + "_embedding", np.arange(i * K, (i + 1) * K)))
        base = len(self.categorical) * K
        for j, c in enumerate(self.numeric):
            blocks.append((c.key, np.arange(base + j, base + j + 1)))
        blocks.sort(key=lambda b: b[0])
        return np.concatenate([b[1] for b in blocks])

    def _tf_variables(self) -> Dict[str, np.ndarray]:
        e = self._engine
        p = e.get_params()
        out: Dict[str, np.ndarray] = {}
        if "linear" in p:
            for i, c in enumerate(self.categorical):
                out["linear/linear_model/%s/weights" % c.key] = p["linear"][self.offsets[i]:self.offsets[i + 1]].reshape(-1, 1)
            for j, c in enumerate(self.numeric):
                out["linear/linear_model/%s/weights" % c.key] = p["linear_dense"][j].reshape(1, 1)
            out["linear/linear_model/bias_weights"] = p["bias"]
        if "emb" in p:
            for i, c in enumerate(self.categorical):
                out["dnn/input_from_feature_columns/input_layer/%s_embedding/embedding_weights" % c.key] = \
                    p["emb"][self.offsets[i]:self.offsets[i + 1]]
            order = self._dnn_input_order()
            for i in range(len(self.hidden_units)):
                w = p["mlp%d/weights" % i]
                out["dnn/hiddenlayer_%d/kernel" % i] = w[order] if i == 0 else w
                out["dnn/hiddenlayer_%d/bias" % i] = p["mlp%d/biases" % i]
            out["dnn/logits/kernel"] = p["deep_out/weights"]
            out["dnn/logits/bias"] = p["deep_out/biases"]
        return out

    def _set_tf_variables(self, tfv: Dict[str, np.ndarray]) -> None:
        e = self._engine
        shapes = e.param_shapes
        if "linear" in shapes:
            lin = np.concatenate([np.asarray(tfv["linear/linear_model/%s/weights" % c.key]).reshape(-1) for c in self.categorical])
            e.set_param("linear", lin.astype(np.float32))
            e.set_param("linear_dense", np.asarray([np.asarray(tfv["linear/linear_model/%s/weights" % c.key]).reshape(())
                                                    for c in self.numeric], dtype=np.float32))
            e.set_param("bias", np.asarray(tfv["linear/linear_model/bias_weights"], dtype=np.float32).reshape(1))
        if "emb" in shapes:
            emb = np.concatenate([np.asarray(tfv["dnn/input_from_feature_columns/input_layer/%s_embedding/embedding_weights" % c.key])
                                  for c in self.categorical])
            e.set_param("emb", emb.astype(np.float32))
            order = self._dnn_input_order()
            for i in range(len(self.hidden_units)):
                w = np.asarray(tfv["dnn/hiddenlayer_%d/kernel" % i], dtype=np.float32)
                if i == 0:
                    back = np.empty_like(w)
                    back[order] = w
                    w = back
                e.set_param("mlp%d/weights" % i, w)
                e.set_param("mlp%d/biases" % i, np.asarray(tfv["dnn/hiddenlayer_%d/bias" % i], dtype=np.float32))
            e.set_param("deep_out/weights", np.asarray(tfv["dnn/logits/kernel"], dtype=np.float32))
            e.set_param("deep_out/biases", np.asarray(tfv["dnn/logits/bias"], dtype=np.float32).reshape(1))

    def get_variable_names(self):
        return sorted(self._tf_variables()) if self._engine is not None else []

    def get_variable_value(self, name):
        return self._tf_variables()[name]

    # -- checkpoints: engine-layout arrays + optimizer slots (resume), TF-named variables go to export ---------------------------
    def _snapshot(self) -> Dict[str, np.ndarray]:
        e = self._engine
        out = {"global_step": np.int64(e.global_step)}
        for name in e.param_shapes:
            out[name] = e.get_param(name)
            out[name + "/slot0"] = e.get_slot(name, 0)
            out[name + "/slot1"] = e.get_slot(name, 1)
        return out

    def _restore(self, ck) -> None:
        e = self._engine
        for name, shp in e.param_shapes.items():
            if name not in ck:
                raise errors.NotFoundError("Key %s not found in checkpoint" % name)
            e.set_param(name, np.asarray(ck[name]).reshape(shp))
            if name + "/slot0" in ck:
                e.set_slot(name, 0, np.asarray(ck[name + "/slot0"]).reshape(shp))
                e.set_slot(name, 1, np.asarray(ck[name + "/slot1"]).reshape(shp))
        e.global_step = int(ck["global_step"]) if "global_step" in ck else 0

    def latest_checkpoint(self) -> Optional[str]:
        files = glob.glob(os.path.join(self.model_dir, "model.ckpt-*.npz"))
        return max(files, key=lambda f: int(f.rsplit("-", 1)[1].split(".")[0])) if files else None

    def _load_latest(self):
        f = self.latest_checkpoint()
        return dict(np.load(f)) if f else None

    def _save(self) -> str:
        os.makedirs(self.model_dir, exist_ok=True)
        snap = self._snapshot()
        path = os.path.join(self.model_dir, "model.ckpt-%d.npz" % int(snap["global_step"]))
        np.savez(path, **snap)
        return path

    # -- input ------------------------------------------------------------------------------------------------------------------
    def _pipeline(self, input_fn) -> D.Dataset:
        with G.Graph() as g:
            out = input_fn()
            if isinstance(out, D.Dataset):
                out = out.make_one_shot_iterator().get_next()
            feats = out[0] if isinstance(out, tuple) else out
            ds = g.collections.get("iterators", [None])[-1]
        if ds is None or ds.csv is None:
            raise errors.InvalidArgumentError("input_fn must return the (features, labels) of a decode_csv tf.data pipeline")
        missing = [c.key for c in self.numeric + self.categorical if c.key not in feats]
        if missing:
            raise ValueError("Feature %s is not in features dictionary." % missing[0])
        return ds

    def _device_batches(self, ds: D.Dataset):
        """-> (rows i32 [b,Fc], ones f32 [b,Fc], numeric f32 [b,Nd], labels f32 [b]) on the GPU"""
        import torch
        dev = torch.device("cuda", torch.cuda.current_device())
        fn, inames = ds.csv_float_names, ds.csv_int_names
        num_idx = [fn.index(c.key) for c in self.numeric]
        cat_idx = [inames.index(c.key) for c in self.categorical]
        lab_idx = fn.index("__label__")
        buckets = np.asarray([c.num_buckets for c in self.categorical], dtype=np.int64)[None, :]
        defaults = np.asarray([0 if c.default_value is None else c.default_value for c in self.categorical], dtype=np.int64)[None, :]
        ones = None
        for f, i in ds.csv_batches():
            cat = i[:, cat_idx].astype(np.int64)
            bad = (cat < 0) | (cat >= buckets)
            if bad.any():
                for c, col in enumerate(self.categorical):
                    if col.default_value is None and bad[:, c].any():
                        raise errors.InvalidArgumentError("%s: id %d is outside [0, %d) and the column has no default_value"
                                                          % (col.key, int(cat[bad[:, c], c][0]), col.num_buckets))
                cat = np.where(bad, defaults, cat)
            rows = (cat + self.offsets[None, :-1]).astype(np.int32)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory().to(dev, non_blocking=True)
            if ones is None or ones.shape[0] != rows.shape[0]:
                ones = torch.ones(rows.shape[0], rows.shape[1], dtype=torch.float32, device=dev)
            yield t(rows), ones, t(f[:, num_idx]), t(f[:, lab_idx])

    # -- modes ------------------------------------------------------------------------------------------------------------------
    def train(self, input_fn, hooks=None, steps=None, max_steps=None, saving_listeners=None):
        from . import logging as L
        ds = self._pipeline(input_fn)
        e = self._ensure_engine(ds.batch_size)
        log_every = max(1, int(self._config.log_step_count_steps or 100))
        start = e.global_step
        done, t0, n0 = 0, time.time(), 0
        for rows, ones, numeric, labels in self._device_batches(ds):
            if (steps is not None and done >= steps) or (max_steps is not None and start + done >= max_steps):
                break
            want = (done + 1) % log_every == 0
            loss = e.train_step(rows, ones, labels, want_loss=want, dense=numeric)
            done += 1
            n0 += int(labels.shape[0])
            if want:
                dt = time.time() - t0
                L.info("global_step/sec: %.4g  examples/sec: %.4g  loss = %.7g, step = %d" % (log_every / dt, n0 / dt, loss, start + done))
                t0, n0 = time.time(), 0
        e.check_ids()
        L.info("Saving checkpoints for %d into %s." % (e.global_step, self._save()))
        return self

    def evaluate(self, input_fn, steps=None, hooks=None, checkpoint_path=None, name=None):
        from . import logging as L
        ds = self._pipeline(input_fn)
        e = self._ensure_engine(ds.batch_size)
        e.eval_reset()
        n = 0
        for rows, ones, numeric, labels in self._device_batches(ds):
            if steps is not None and n >= steps:
                break
            e.eval_batch(rows, ones, labels, dense=numeric)
            n += 1
        auc, avg_loss, count = e.eval_result()
        # canned binary head: average_loss = mean per example, loss = mean per batch of the batch sums [TF-1.4 head metrics]
        out = {"auc": auc, "average_loss": avg_loss, "loss": avg_loss * count / max(n, 1), "global_step": e.global_step}
        L.info("Saving dict for global step %d: %s" % (e.global_step, ", ".join("%s = %s" % kv for kv in sorted(out.items()))))
        return out

    def predict(self, input_fn, predict_keys=None, hooks=None, checkpoint_path=None, yield_single_examples=True) -> Iterator[Dict[str, Any]]:
        import torch
        ds = self._pipeline(input_fn)
        e = self._ensure_engine(ds.batch_size)
        want = None if predict_keys is None else ([predict_keys] if isinstance(predict_keys, str) else list(predict_keys))
        for rows, ones, numeric, _labels in self._device_batches(ds):
            b = int(rows.shape[0])
            prob = torch.empty(b, dtype=torch.float32, device=rows.device)
            logit = torch.empty(b, dtype=torch.float32, device=rows.device)
            e.predict(rows, ones, prob, logit, dense=numeric)
            p, y = prob.cpu().numpy(), logit.cpu().numpy()
            full = {"logits": y[:, None], "logistic": p[:, None], "probabilities": np.stack([1.0 - p, p], axis=1),
                    "class_ids": (p > 0.5).astype(np.int64)[:, None],
                    "classes": np.asarray([[b"1"] if v > 0.5 else [b"0"] for v in p], dtype=object)}
            keys = [k for k in full if want is None or k in want]
            if yield_single_examples:
                for r in range(b):
                    yield {k: full[k][r] for k in keys}
            else:
                yield {k: full[k] for k in keys}

    def export_savedmodel(self, export_dir_base, serving_input_receiver_fn, assets_extra=None, as_text=False, checkpoint_path=None, **_kw):
        """Variables under their TF-1.4 names + the parsing signature (tf.Example with `feature_spec`); not a TF protobuf."""
        recv = serving_input_receiver_fn()
        self._ensure_engine(1024)
        out = os.path.join(export_dir_base, str(int(time.time())))
        os.makedirs(out, exist_ok=True)
        np.savez(os.path.join(out, "variables.npz"), **self._tf_variables())
        sig = {"signature_def": {"serving_default": {
            "inputs": {"examples": {"dtype": "string", "shape": [None]}},
            "feature_spec": getattr(recv, "feature_spec", None),
            "outputs": {"classes": {"dtype": "string", "shape": [None, 2]}, "scores": {"dtype": "float32", "shape": [None, 2]}},
            "method_name": "tensorflow/serving/classify"}},
            "engine": {"model": self.model_type, "categorical": [[c.key, c.num_buckets, c.default_value] for c in self.categorical],
                       "numeric": [c.key for c in self.numeric], "embedding_size": self.dimension, "hidden_units": self.hidden_units}}
        with open(os.path.join(out, "signature.json"), "w") as f:
            json.dump(sig, f, indent=1)
        return out

    def close(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None


class LinearClassifier(_CannedEstimator):
    def __init__(self, feature_columns, model_dir=None, n_classes=2, weight_column=None, label_vocabulary=None, optimizer="Ftrl",
                 config=None, partitioner=None):
        if optimizer not in ("Ftrl", None) or weight_column is not None:
            raise errors.UnimplementedError("LinearClassifier: only the default Ftrl optimizer, no weight_column")
        super().__init__("wide", model_dir, feature_columns, None, None, config, n_classes)


class DNNClassifier(_CannedEstimator):
    def __init__(self, hidden_units, feature_columns, model_dir=None, n_classes=2, weight_column=None, label_vocabulary=None,
                 optimizer="Adagrad", activation_fn=None, dropout=None, input_layer_partitioner=None, config=None):
        if optimizer not in ("Adagrad", None) or weight_column is not None or dropout is not None or activation_fn is not None:
            raise errors.UnimplementedError("DNNClassifier: only the defaults (Adagrad, relu, no dropout, no weight_column)")
        super().__init__("deep", model_dir, None, feature_columns, list(hidden_units), config, n_classes)


class DNNLinearCombinedClassifier(_CannedEstimator):
    def __init__(self, model_dir=None, linear_feature_columns=None, linear_optimizer="Ftrl", dnn_feature_columns=None,
                 dnn_optimizer="Adagrad", dnn_hidden_units=None, dnn_activation_fn=None, dnn_dropout=None, n_classes=2,
                 weight_column=None, label_vocabulary=None, input_layer_partitioner=None, config=None):
        if linear_optimizer not in ("Ftrl", None) or dnn_optimizer not in ("Adagrad", None) or dnn_dropout is not None \
                or dnn_activation_fn is not None or weight_column is not None:
            raise errors.UnimplementedError("DNNLinearCombinedClassifier: only the defaults (Ftrl / Adagrad, relu, no dropout)")
        if not linear_feature_columns or not dnn_feature_columns:
            raise ValueError("Either linear_feature_columns or dnn_feature_columns must be defined.")
        super().__init__("wide_n_deep", model_dir, linear_feature_columns, dnn_feature_columns, list(dnn_hidden_units or []), config,
                         n_classes)
