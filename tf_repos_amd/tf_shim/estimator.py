"""tf.estimator surface the reference drives (DeepFM.py:339-366): Estimator(model_fn, model_dir, params, config),
train / evaluate / predict / export_savedmodel, TrainSpec / EvalSpec / train_and_evaluate, RunConfig, EstimatorSpec.

model_fn and input_fn are called exactly as TF would call them (graph mode); the graph they build is lowered onto the
HIP engine once per mode, then the Python loop only moves batches: host parser -> pinned memory -> GPU -> one engine
call per step.  Checkpoints are .npz files keyed by the TF variable names in model_dir.
"""
from __future__ import annotations

import glob
import inspect
import json
import os
import time
from typing import Any, Callable, Dict, Iterator, List, Optional

import numpy as np

from .. import errors
from ..engine import Engine
from . import data as D
from . import graph as G
from .lowering import Lowered, lower


class ModeKeys:
    TRAIN = "train"
    EVAL = "eval"
    PREDICT = "infer"


class EstimatorSpec:
    def __init__(self, mode, predictions=None, loss=None, train_op=None, eval_metric_ops=None, export_outputs=None, **_kw):
        self.mode, self.predictions, self.loss, self.train_op = mode, predictions, loss, train_op
        self.eval_metric_ops, self.export_outputs = eval_metric_ops or {}, export_outputs or {}
        if mode == ModeKeys.TRAIN and (loss is None or train_op is None):
            raise ValueError("Missing loss/train_op for mode TRAIN")
        if mode == ModeKeys.EVAL and loss is None:
            raise ValueError("Missing loss for mode EVAL")
        if mode == ModeKeys.PREDICT and predictions is None:
            raise ValueError("Missing predictions for mode PREDICT")


class ConfigProto:
    def __init__(self, device_count=None, **kw):
        self.device_count = device_count or {}
        self.__dict__.update(kw)


class RunConfig:
    def __init__(self, model_dir=None, session_config=None, log_step_count_steps=100, save_summary_steps=100,
                 save_checkpoints_steps=None, save_checkpoints_secs=600, keep_checkpoint_max=5, tf_random_seed=None, **_kw):
        self.model_dir = model_dir
        self.session_config = session_config
        self.log_step_count_steps = log_step_count_steps
        self.save_summary_steps = save_summary_steps
        self.save_checkpoints_steps = save_checkpoints_steps
        self.save_checkpoints_secs = save_checkpoints_secs
        self.keep_checkpoint_max = keep_checkpoint_max
        self.tf_random_seed = tf_random_seed

    def replace(self, **kw):
        c = RunConfig(**{k: v for k, v in self.__dict__.items()})
        for k, v in kw.items():
            if not hasattr(c, k):
                raise ValueError("Replacing %s is not supported" % k)
            setattr(c, k, v)
        return c


class TrainSpec:
    def __init__(self, input_fn, max_steps=None, hooks=None):
        self.input_fn, self.max_steps = input_fn, max_steps


class EvalSpec:
    def __init__(self, input_fn, steps=100, name=None, hooks=None, exporters=None, start_delay_secs=120, throttle_secs=600):
        self.input_fn, self.steps = input_fn, steps
        self.start_delay_secs, self.throttle_secs = start_delay_secs, throttle_secs


class PredictOutput:
    def __init__(self, outputs):
        self.outputs = outputs


class ServingInputReceiver:
    def __init__(self, features, receiver_tensors):
        self.features, self.receiver_tensors = features, receiver_tensors


def build_raw_serving_input_receiver_fn(features, default_batch_size=None):
    def fn():
        return ServingInputReceiver(dict(features), dict(features))
    return fn


def _init_value(v: G.Variable, rng: np.random.Generator) -> np.ndarray:
    """Initial values per SURVEY Appendix B items 4, 10 [TF-1.4]: glorot_normal = truncated normal (resample beyond 2
    sigma), sigma = sqrt(2/(fan_in+fan_out)); rank-1 [V] -> fan_in = fan_out = V; glorot_uniform U(+-sqrt(6/(fi+fo)))."""
    shp = v.shape
    init = v.initializer
    if init.kind == "constant":
        return np.full(shp, init.kw["value"], dtype=np.float32)
    if len(shp) == 1:
        fi = fo = shp[0]
    else:
        fi, fo = int(np.prod(shp[:-1])), shp[-1]
    if init.kind == "glorot_uniform":
        lim = np.sqrt(6.0 / (fi + fo))
        return rng.uniform(-lim, lim, size=shp).astype(np.float32)
    if init.kind == "glorot_normal":
        std = np.sqrt(2.0 / (fi + fo)) / 0.87962566103423978      # TF's truncated-normal variance correction
        a = rng.normal(0, std, size=shp)
        bad = np.abs(a) > 2 * std
        while bad.any():
            a[bad] = rng.normal(0, std, size=int(bad.sum()))
            bad = np.abs(a) > 2 * std
        return a.astype(np.float32)
    raise errors.UnimplementedError("initializer %s" % init.kind)


def _norm(v):
    return tuple(v) if isinstance(v, (list, tuple)) else v


class Estimator:
    def __init__(self, model_fn=None, model_dir=None, config=None, params=None, warm_start_from=None):
        self._model_fn = model_fn
        self._config = config or RunConfig()
        self.model_dir = model_dir or self._config.model_dir or "/tmp/tf_repos_amd_model"
        self.params = dict(params or {})
        self._engine: Optional[Engine] = None
        self._lowered: Optional[Lowered] = None
        self._variables: Dict[str, G.Variable] = {}
        self.table_mode = os.environ.get("DCTR_TABLE_MODE", "dense_exact")

    @property
    def config(self):
        return self._config

    # -- graph construction --------------------------------------------------------------------------------------------
    def _call_model_fn(self, features, labels, mode):
        args = inspect.signature(self._model_fn).parameters
        kw = {}
        if "labels" in args:
            kw["labels"] = labels
        if "mode" in args:
            kw["mode"] = mode
        if "params" in args:
            kw["params"] = self.params
        if "config" in args:
            kw["config"] = self._config
        return self._model_fn(features=features, **kw)

    def _build(self, input_fn, mode):
        from . import FLAGS_MODULE
        with G.Graph() as g:
            out = input_fn()
            if isinstance(out, D.Dataset):
                out = out.make_one_shot_iterator().get_next()
            features, labels = out if isinstance(out, tuple) else (out, None)
            if mode == ModeKeys.PREDICT:
                labels = None
            spec = self._call_model_fn(features, labels, mode)
            if not isinstance(spec, EstimatorSpec):
                raise ValueError("model_fn should return an EstimatorSpec.")
            lowered = lower(spec.loss, spec.train_op, spec.predictions or {}, FLAGS_MODULE.FLAGS)
            pipeline = g.collections.get("iterators", [None])[-1]
            if pipeline is not None and lowered.slots is not None:
                from ..tfrecord import SlotSpec
                pipeline.slot_specs = [SlotSpec(*s) for s in lowered.slots]   # the concat order of the model fixes the slot layout
                pipeline.label_keys = list(lowered.label_keys)
                pipeline.feature_size = int(lowered.config_kwargs["feature_size"])
            elif pipeline is not None:
                pipeline.field_size = lowered.config_kwargs["field_size"]     # reshape(feat_ids, [-1, field_size]) fixes F
            variables = dict(g.variables)
        return spec, lowered, pipeline, variables

    # hyper-parameters only a TRAIN graph carries (an EVAL / PREDICT graph has no train_op and no dropout, so its lowering falls
    # back to EngineConfig's defaults for them)
    _TRAIN_ONLY = ("optimizer", "learning_rate", "dropout", "lin_optimizer", "lin_learning_rate")

    def _ensure_engine(self, lowered: Lowered, variables, batch_size: int, max_entries: int = 0, mode: str = ModeKeys.TRAIN):
        """Capacity (max_batch / max_entries) and hyper-parameters are separate concerns.  The engine is always configured from the
        TRAIN lowering once one has been seen: evaluate() / predict() before train(), or an evaluate() with a larger batch in
        between two train() calls, must not leave training running with Adam@5e-4 and no dropout.  A TRAIN lowering that
        disagrees with the live engine rebuilds it (variables kept; optimizer slots kept when the optimizer is the same)."""
        if mode == ModeKeys.TRAIN:
            self._train_lowered = lowered
        cfg_src = getattr(self, "_train_lowered", None) or lowered
        live = self._engine
        stale = False
        if live is not None and mode == ModeKeys.TRAIN:
            kw = cfg_src.config_kwargs
            stale = any(k in kw and _norm(kw[k]) != _norm(getattr(live.cfg, k)) for k in self._TRAIN_ONLY)
        need_new = live is None or stale or live.cfg.max_batch < batch_size or live.cfg.max_entries < max_entries
        if not need_new:
            return
        state = self._snapshot() if live is not None else None
        if live is not None:
            if stale and _norm(cfg_src.config_kwargs.get("optimizer")) != _norm(live.cfg.optimizer) and state is not None:
                # slots of another optimizer mean nothing.  An engine built by evaluate() / predict() has not trained: what it holds
                # came from the latest checkpoint (whose slots belong to the optimizer that wrote it) or from the initializers
                disk = None if getattr(self, "_trained_since_build", False) else self._load_latest()
                state = disk if disk is not None else {k: v for k, v in state.items() if not k.endswith(("/slot0", "/slot1"))}
            batch_size = max(batch_size, live.cfg.max_batch)
            max_entries = max(max_entries, live.cfg.max_entries)
            live.close()
        extra = {"max_entries": int(max_entries)} if cfg_src.slots is not None else {}
        if cfg_src.slots is None:
            # Eager steps at every batch size.  Rounds 2-3 replayed one captured hipGraph per (batch size, input slot) for small
            # batches (the reference's default 256), when enqueueing ~25 launches one by one cost this loop 0.2-0.3 ms per step.  Since
            # the loop runs on the engine's own stream with the slot handshake inside the library it costs ~0.13 ms, and the eager step
            # keeps what a captured one cannot have: lagging table rows (csrc/lag.h) and the next-batch grouping hint.  Measured
            # through this Estimator at BASELINE configs[0] (B = 256, V = 117 581, K = 8; tools/e2e_probe.py c1, same box): replayed
            # 0.186-0.189 ms per step = 1.36 M examples/s, eager 0.136-0.148 = 1.72-1.88 M.  A/B knob DCTR_EST_GRAPH=1: the replay.
            # (decided ONCE per Estimator: a later evaluate() must not flip the training path between rebuilds)
            if getattr(self, "_use_graph", None) is None:
                self._use_graph = os.environ.get("DCTR_EST_GRAPH") == "1"
            extra["use_graph"] = self._use_graph
        cfg = cfg_src.engine_config(max_batch=batch_size, table_mode=self.table_mode,
                                    seed=int(self._config.tf_random_seed or 0), **extra)
        self._engine = Engine(cfg)
        self._lowered = cfg_src
        self._variables = variables
        self._trained_since_build = False
        ck = state or self._load_latest()
        if ck is not None:
            self._restore(ck)
        else:
            rng = np.random.default_rng(int(self._config.tf_random_seed or 0))
            for ename, tfname in cfg_src.name_map.items():
                self._engine.set_param(ename, _init_value(variables[tfname], rng).reshape(self._engine.param_shapes[ename]))

    # -- checkpoints (TF variable names) ----------------------------------------------------------------------------------
    def _snapshot(self) -> Dict[str, np.ndarray]:
        e, m = self._engine, self._lowered.name_map
        out = {"global_step": np.int64(e.global_step)}
        for ename, tfname in m.items():
            out[tfname] = e.get_param(ename)
            out[tfname + "/slot0"] = e.get_slot(ename, 0)
            out[tfname + "/slot1"] = e.get_slot(ename, 1)
        return out

    def _restore(self, ck) -> None:
        e, m = self._engine, self._lowered.name_map
        for ename, tfname in m.items():
            if tfname not in ck:
                raise errors.NotFoundError("Key %s not found in checkpoint" % tfname)
            e.set_param(ename, np.asarray(ck[tfname]).reshape(e.param_shapes[ename]))
            for which in (0, 1):
                if tfname + "/slot%d" % which in ck:
                    e.set_slot(ename, which, np.asarray(ck[tfname + "/slot%d" % which]).reshape(e.param_shapes[ename]))
        e.global_step = int(ck["global_step"]) if "global_step" in ck else 0

    def latest_checkpoint(self) -> Optional[str]:
        files = [f for f in glob.glob(os.path.join(self.model_dir, "model.ckpt-*.npz")) if ".tmp." not in f]
        if not files:
            return None
        return max(files, key=lambda f: int(f.rsplit("-", 1)[1].split(".")[0]))

    def _load_latest(self):
        """The newest checkpoint of model_dir: this package's .npz, else a TensorFlow checkpoint bundle (model.ckpt-N.index +
        .data-*), e.g. one a TF run of the reference left there -- read by variable name (tf_bundle.py)."""
        f = self.latest_checkpoint()
        if f:
            return dict(np.load(f))
        from .. import tf_bundle
        prefix = tf_bundle.latest_tf_checkpoint(self.model_dir)
        if prefix is None:
            return None
        opt = (getattr(self, "_train_lowered", None) or self._lowered).config_kwargs.get("optimizer", "Adam") if (getattr(self, "_train_lowered", None) or self._lowered) else "Adam"
        return tf_bundle.bundle_to_state(tf_bundle.read_bundle(prefix), opt)

    def export_tf_checkpoint(self, prefix: Optional[str] = None) -> str:
        """Writes the live variables (and optimizer slots, under TF's slot names) as a TensorFlow checkpoint bundle that
        tf.train.Saver / tf.estimator can restore; returns the prefix."""
        from .. import tf_bundle
        snap = self._snapshot()
        prefix = prefix or os.path.join(self.model_dir, "model.ckpt-%d" % int(snap["global_step"]))
        tf_bundle.write_bundle(prefix, tf_bundle.state_to_bundle(snap, self._lowered.config_kwargs.get("optimizer", "Adam")))
        with open(os.path.join(os.path.dirname(prefix), "checkpoint"), "w") as f:
            f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (os.path.basename(prefix), os.path.basename(prefix)))
        return prefix

    def _save(self) -> str:
        os.makedirs(self.model_dir, exist_ok=True)
        snap = self._snapshot()
        path = os.path.join(self.model_dir, "model.ckpt-%d.npz" % int(snap["global_step"]))
        tmp = path + ".tmp.%d.npz" % os.getpid()       # a crash mid-save must not leave a corrupt file that wins latest_checkpoint()
        np.savez(tmp, **snap)
        os.replace(tmp, path)
        keep = self._config.keep_checkpoint_max or 5
        files = sorted((f for f in glob.glob(os.path.join(self.model_dir, "model.ckpt-*.npz")) if ".tmp." not in f),
                       key=lambda f: int(f.rsplit("-", 1)[1].split(".")[0]))
        for f in files[:-keep]:
            os.remove(f)
        return path

    def get_variable_names(self):
        return sorted(self._lowered.name_map.values()) if self._lowered else []

    def get_variable_value(self, name):
        inv = {v: k for k, v in self._lowered.name_map.items()}
        return self._engine.get_param(inv[name])

    # -- modes ----------------------------------------------------------------------------------------------------------------
    def _device_batches(self, pipeline: "D.Dataset"):
        import torch
        dev = torch.device("cuda", torch.cuda.current_device())
        for ids, vals, labels in pipeline.numpy_batches():
            yield (torch.from_numpy(np.ascontiguousarray(ids)).pin_memory().to(dev, non_blocking=True),
                   torch.from_numpy(np.ascontiguousarray(vals)).pin_memory().to(dev, non_blocking=True),
                   torch.from_numpy(np.ascontiguousarray(labels)).pin_memory().to(dev, non_blocking=True))

    # CSR (multi-hot) models: TFRecord pipelines in the slot layout the model was lowered to
    def _csr_batches(self, pipeline: "D.Dataset"):
        import torch
        dev = torch.device("cuda", torch.cuda.current_device())
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory().to(dev, non_blocking=True)
        for off, ids, wts, labels in pipeline.slot_batches():
            yield up(off), up(ids), up(wts), up(labels[0]), (up(labels[1]) if labels.shape[0] > 1 else None)

    @staticmethod
    def _csr_capacity(pipeline: "D.Dataset") -> int:
        """an upper bound of any batch's entry count: batch_size x the largest example"""
        from ..tfrecord import TFRecordSlotDataset
        if not pipeline.filenames:
            return 0
        ds = TFRecordSlotDataset(pipeline.filenames, pipeline.slot_specs, pipeline.label_keys, pipeline.feature_size)
        off = ds._load()[0]
        per_example = np.diff(off[::ds.n_slots])
        return int(pipeline.batch_size * (per_example.max() if len(per_example) else 0))

    def _metric_outputs(self, spec, lowered) -> Dict[str, int]:
        """eval metric key -> engine output index (its tf.metrics.auc reads one of the predictions)"""
        by_id = {id(t): lowered.outputs.get(k, 0) for k, t in (spec.predictions or {}).items()}
        out = {}
        for key, val in (spec.eval_metric_ops or {}).items():
            node = val[0] if isinstance(val, tuple) else val
            out[key] = by_id.get(id(node.inputs[1]), 0) if node is not None and node.op == "metrics_auc" else 0
        return out

    def train(self, input_fn, hooks=None, steps=None, max_steps=None, saving_listeners=None):
        from . import logging as L
        spec, lowered, pipeline, variables = self._build(input_fn, ModeKeys.TRAIN)
        if pipeline is None:
            raise errors.InvalidArgumentError("input_fn must return tensors produced by a tf.data iterator")
        csr = lowered.slots is not None
        self._ensure_engine(lowered, variables, pipeline.batch_size, self._csr_capacity(pipeline) if csr else 0, ModeKeys.TRAIN)
        self._trained_since_build = True
        e = self._engine
        log_every = max(1, int(self._config.log_step_count_steps or 100))
        start_step = e.global_step
        t0, n0 = time.time(), 0
        done = 0
        loss = None
        # fixed-field libsvm pipelines: batches are staged into the engine's input slots by a background feeder
        feeder = None
        side = None
        if not csr and pipeline.csv is None and not getattr(e.cfg, "dense_size", 0):
            from ..feeder import DeviceFeeder
            import torch
            # The feeder-driven loop runs on the ENGINE'S OWN stream (dctr_main_stream), not on the legacy default stream: there every
            # record / wait that ties the steps to the feeder's copies costs more (tools/feeder_breakdown.py: 312 vs 272 us per step at
            # c2).  A/B knob DCTR_EST_MAIN_STREAM=0: the default stream (round 3).
            if os.environ.get("DCTR_EST_MAIN_STREAM", "1") == "1":
                torch.cuda.synchronize()              # (what built the engine and loaded its variables is complete)
                side = self._side_stream = e.main_stream()
                self._side_ctx = torch.cuda.stream(side)
                self._side_ctx.__enter__()
            feeder = DeviceFeeder(e, pipeline.numpy_batches())
        try:
            self._train_loop(e, csr, feeder, pipeline, steps, max_steps, start_step, log_every)
            if feeder is not None:                    # (before the id check and the checkpoint save: no H2D copy into an input slot, no pending
                feeder.close()                        #  next-batch grouping may run beside them)
                feeder = None
            e.check_ids()
            path = self._save()
            from . import logging as L
            L.info("Saving checkpoints for %d into %s." % (e.global_step, path))
            return self
        finally:
            if feeder is not None:                    # (also on an exception in train_step / check_ids: the thread stops, its pending hint is dropped)
                feeder.close()
            if side is not None:
                side.synchronize()
                self._side_ctx.__exit__(None, None, None)

    def _train_loop(self, e, csr, feeder, pipeline, steps, max_steps, start_step, log_every):
        from . import logging as L
        t0, n0 = time.time(), 0
        done = 0
        loss = None
        for batch in (self._csr_batches(pipeline) if csr else (feeder if feeder is not None else self._device_batches(pipeline))):
            if steps is not None and done >= steps:
                break
            if max_steps is not None and start_step + done >= max_steps:
                break
            want = (done + 1) % log_every == 0
            if csr:
                off, ids, wts, labels, z = batch
                loss = e.train_step_csr(off, ids, wts, labels, z, want_loss=want)
            elif feeder is not None:
                ids, vals, labels, slot = batch
                loss = e.train_step(ids, vals, labels, want_loss=want)
                feeder.release(slot)
                if not e.cfg.use_graph:                # (a replayed graph groups the ids inside the step)
                    nxt = feeder.peek_next_ids(wait=0.05)      # the next batch, staged (or being staged) in its input slot: group its ids a step ahead
                    if nxt is not None:
                        e.prefetch_ids(nxt)
            else:
                ids, vals, labels = batch
                loss = e.train_step(ids, vals, labels, want_loss=want)
            done += 1
            n0 += int(labels.shape[0])
            if want:
                e.check_ids()
                dt = time.time() - t0
                L.info("global_step/sec: %.4g  examples/sec: %.4g  loss = %.7g, step = %d" % (log_every / dt, n0 / dt, loss, start_step + done))
                t0, n0 = time.time(), 0
        return self

    def evaluate(self, input_fn, steps=None, hooks=None, checkpoint_path=None, name=None):
        from . import logging as L
        spec, lowered, pipeline, variables = self._build(input_fn, ModeKeys.EVAL)
        csr = lowered.slots is not None
        self._ensure_engine(lowered, variables, pipeline.batch_size, self._csr_capacity(pipeline) if csr else 0, ModeKeys.EVAL)
        e = self._engine
        e.eval_reset()
        n = 0
        for batch in (self._csr_batches(pipeline) if csr else self._device_batches(pipeline)):
            if steps is not None and n >= steps:
                break
            if csr:
                e.eval_batch_csr(*batch)
            else:
                e.eval_batch(*batch)
            n += 1
        e.check_ids()
        auc, loss, _count = e.eval_result()
        out = {"loss": loss, "global_step": e.global_step}
        which = self._metric_outputs(spec, lowered) if csr else {}
        for key in (spec.eval_metric_ops or {"auc": None}):
            out[key] = e.eval_auc_extra(which[key]) if which.get(key, 0) else auc
        L.info("Saving dict for global step %d: %s" % (e.global_step, ", ".join("%s = %s" % kv for kv in sorted(out.items()))))
        return out

    def predict(self, input_fn, predict_keys=None, hooks=None, checkpoint_path=None, yield_single_examples=True) -> Iterator[Dict[str, Any]]:
        import torch
        spec, lowered, pipeline, variables = self._build(input_fn, ModeKeys.PREDICT)
        csr = lowered.slots is not None
        self._ensure_engine(lowered, variables, pipeline.batch_size, self._csr_capacity(pipeline) if csr else 0, ModeKeys.PREDICT)
        e = self._engine
        keys = list(spec.predictions.keys())
        if predict_keys is not None:
            want = [predict_keys] if isinstance(predict_keys, str) else list(predict_keys)
            keys = [k for k in keys if k in want]
        if csr:
            for off, ids, wts, y, _z in self._csr_batches(pipeline):
                B = int(y.shape[0])
                outs = [torch.empty(B, dtype=torch.float32, device=ids.device) for _ in range(3)]
                e.predict_csr(off, ids, wts, B, *outs)
                host = [o.cpu().numpy() for o in outs]
                e.check_ids()
                if yield_single_examples:
                    for i in range(B):
                        yield {k: host[lowered.outputs[k]][i] for k in keys}
                else:
                    yield {k: host[lowered.outputs[k]] for k in keys}
            return
        for ids, vals, _labels in self._device_batches(pipeline):
            prob = torch.empty(int(ids.shape[0]), dtype=torch.float32, device=ids.device)
            e.predict(ids, vals, prob, None)
            p = prob.cpu().numpy()
            e.check_ids()
            if yield_single_examples:
                for v in p:
                    yield {k: v for k in keys}
            else:
                yield {k: p for k in keys}

    def export_savedmodel(self, export_dir_base, serving_input_receiver_fn, assets_extra=None, as_text=False, checkpoint_path=None, **_kw):
        """Writes the variables (TF names) and the serving signature: inputs feat_ids int64 [None,F] / feat_vals float32
        [None,F], output key(s) of `predictions` (DeepFM.py:361-366).  Not a TF SavedModel protobuf (no TF here)."""
        recv = serving_input_receiver_fn()
        if self._lowered is not None and self._lowered.slots is not None:
            raise errors.UnimplementedError("export of the DIN / ESMM models: the scripts' own serving spec (feat_ids / feat_vals only, "
                                            "DIN.py:387-391) does not feed their model_fn")
        with G.Graph() as g:
            feats = {k: G.placeholder(v.dtype, v.shape, name=k) for k, v in recv.features.items()}
            spec = self._call_model_fn(feats, None, ModeKeys.PREDICT)
            from . import FLAGS_MODULE
            lowered = lower(None, None, spec.predictions, FLAGS_MODULE.FLAGS)
            variables = dict(g.variables)
        self._ensure_engine(lowered, variables, 1024, 0, ModeKeys.PREDICT)
        out = os.path.join(export_dir_base, str(int(time.time())))
        os.makedirs(out, exist_ok=True)
        snap = {k: v for k, v in self._snapshot().items() if "/slot" not in k}
        np.savez(os.path.join(out, "variables.npz"), **snap)
        sig = {"signature_def": {"serving_default": {
            "inputs": {k: {"dtype": v.dtype.name, "shape": [None if s is None else int(s) for s in (v.shape or ())]} for k, v in recv.receiver_tensors.items()},
            "outputs": {k: {"dtype": "float32", "shape": [None]} for k in spec.predictions},
            "method_name": "tensorflow/serving/predict"}},
            "engine": {"model": lowered.model, "config": {k: (list(v) if isinstance(v, tuple) else v) for k, v in lowered.config_kwargs.items()},
                       "name_map": lowered.name_map}}
        with open(os.path.join(out, "signature.json"), "w") as f:
            json.dump(sig, f, indent=1)
        return out


def train_and_evaluate(estimator: Estimator, train_spec: TrainSpec, eval_spec: EvalSpec):
    """Local mode of tf.estimator.train_and_evaluate: train until the input is exhausted (or max_steps), then evaluate.
    (TF interleaves evaluations every throttle_secs on the latest checkpoint; the final metrics are the same.)"""
    estimator.train(train_spec.input_fn, max_steps=train_spec.max_steps)
    return estimator.evaluate(eval_spec.input_fn, steps=eval_spec.steps)
