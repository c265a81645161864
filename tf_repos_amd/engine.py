"""Python host for one engine handle (one GPU rank): thin, typed wrapper over the C ABI.

Mirrors what tf.estimator holds for a `model_fn` (DeepFM.py:100-221): the variables, the
optimizer slots, global_step, and the train / predict ops -- but every arithmetic op runs in
libdeepctr_hip.so.  torch tensors are only device buffers whose raw pointers cross the ABI.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import capi, errors


@dataclass
class EngineConfig:
    """Field names follow the reference flags (DeepFM.py:34-60, DCN.py:52, AFM.py:52, PNN.py:61)."""
    model: str = "deepfm"                      # deepfm | fnn | ipnn | opnn | nfm | afm | dcn | mvm | wide | deep | wide_n_deep
    field_size: int = 39
    feature_size: int = 117581
    embedding_size: int = 32
    deep_layers: Sequence[int] = (256, 128, 64)
    dropout: Sequence[float] = (0.5, 0.5, 0.5)  # TF keep_prob
    cross_layers: int = 3
    attention_layers: Sequence[int] = (256,)
    l2_reg: float = 1e-4
    learning_rate: float = 5e-4
    optimizer: str = "Adam"
    table_mode: str = "dense_exact"            # dense_exact (TF semantics) | touched_rows (lazy)
    batch_norm: bool = False
    batch_norm_decay: float = 0.9
    batch_norm_bessel: bool = True             # moving variance fed var * B/(B-1) (TF-1.4's fused batch_norm); False: the biased batch variance
    max_batch: int = 4096
    seed: int = 0
    shard_rank: int = 0
    shard_world: int = 1
    # canned estimators of wide_n_deep.py (model = wide | deep | wide_n_deep): field_size categorical identity columns over
    # one stacked table, dense_size numeric columns as dense inputs, a linear side with its own optimizer, summed loss
    dense_size: int = 0
    lin_optimizer: str = "ftrl"
    lin_learning_rate: float = 0.005
    loss_sum: bool = False
    # CSR (multi-hot) models din | esmm: field_size = number of K-wide slots of the MLP input (DIN.py:199), batches are
    # (offsets [B*S+1], ids [nnz], weights [nnz]) with nnz <= max_entries (0: max_batch * field_size * 8)
    max_entries: int = 0
    ctr_task_wgt: float = 0.5                  # DeepCvrMTL.py:47
    # din with attention pooling (DIN.py:45,151-177): (user multi-hot slot, ad slot) pairs; the attention MLP takes its widths
    # from attention_layers and its keep_probs from dropout[i]
    att_pairs: Sequence[Tuple[int, int]] = ()
    table_sweep_period: int = 0                # dense_exact + Adam: period of the time-blocked table sweep (include/deepctr_hip.h); 0 = default, 1 = classic
    gemm_mode: str = "default"                 # MLP products: "split" (three bf16 planes, six products, f32 accumulate: f32-equivalent results on
                                               # the bf16 matrix pipe), "exact" (f32 MFMA), or "default" = the library's (DCTR_GEMM_MODE, else split;
                                               # include/deepctr_hip.h dctr_config.gemm_mode)
    use_graph: bool = False                    # False: eager launches on 3 HIP streams (measured faster: each stream keeps its own
                                               # hardware queue); True: one captured hipGraph per (batch size, input slot)

    def to_c(self) -> capi.Config:
        if self.model not in capi.MODELS:
            raise errors.InvalidArgumentError("unknown model %r" % self.model)
        if self.optimizer not in capi.OPTIMIZERS:
            # '--optimizer=GD' leaves `optimizer` unbound in the reference (DeepFM.py:204-213)
            raise NameError("name 'optimizer' is not defined (optimizer=%r)" % self.optimizer)
        c = capi.Config()
        c.model = capi.MODELS[self.model]
        c.field_size = self.field_size
        c.embedding_size = self.embedding_size
        c.feature_size = self.feature_size
        layers = list(self.deep_layers)
        keep = list(self.dropout)
        if len(layers) > capi.MAX_LAYERS:
            raise errors.InvalidArgumentError("at most %d deep layers" % capi.MAX_LAYERS)
        c.n_deep_layers = len(layers)
        for i, h in enumerate(layers):
            c.deep_layers[i] = int(h)
        for i in range(capi.MAX_LAYERS):        # AFM reads keep_prob[0..1] = attention / pooled-embedding dropout (AFM.py:153,158)
            c.keep_prob[i] = float(keep[i]) if i < len(keep) else 1.0
        c.cross_layers = self.cross_layers
        att = list(self.attention_layers)
        c.n_attention_layers = len(att)
        for i, a in enumerate(att[:capi.MAX_LAYERS]):
            c.attention_layers[i] = int(a)
        c.l2_reg = self.l2_reg
        c.learning_rate = self.learning_rate
        c.optimizer = capi.OPTIMIZERS[self.optimizer]
        c.table_mode = capi.TABLE_MODES[self.table_mode]
        c.batch_norm = int(self.batch_norm)
        c.batch_norm_decay = self.batch_norm_decay
        c.batch_norm_biased_moving_variance = int(not self.batch_norm_bessel)
        c.max_batch = self.max_batch
        c.seed = self.seed
        c.shard_rank = self.shard_rank
        c.shard_world = self.shard_world
        c.use_graph = int(self.use_graph if os.environ.get("DCTR_FORCE_GRAPH") is None else os.environ["DCTR_FORCE_GRAPH"] == "1")   # (A/B knob)
        c.dense_size = int(self.dense_size)
        c.lin_optimizer = capi.OPTIMIZERS[self.lin_optimizer]
        c.lin_learning_rate = float(self.lin_learning_rate)
        c.loss_sum = int(self.loss_sum)
        c.max_entries = int(self.max_entries)
        c.ctr_task_wgt = float(self.ctr_task_wgt)
        c.table_sweep_period = int(self.table_sweep_period)
        if self.gemm_mode not in ("default", "exact", "split"):
            raise errors.InvalidArgumentError("gemm_mode must be 'default', 'exact' or 'split', got %r" % (self.gemm_mode,))
        c.gemm_mode = {"default": 0, "split": 1, "exact": 2}[self.gemm_mode]
        c.n_att_pairs = len(self.att_pairs)
        for i, (u, a) in enumerate(list(self.att_pairs)[:8]):
            c.att_user_slot[i], c.att_ad_slot[i] = int(u), int(a)
        return c


class Engine:
    def __init__(self, cfg: EngineConfig):
        self.cfg = cfg
        self._lib = capi.lib()
        self._h = C.c_void_p()
        ccfg = cfg.to_c()
        capi.check(self._lib.dctr_create(C.byref(ccfg), C.byref(self._h)))
        self._shapes = self._query_shapes()

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.dctr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- variables ---------------------------------------------------------------------------
    def _query_shapes(self) -> Dict[str, Tuple[int, ...]]:
        n = C.c_int()
        capi.check(self._lib.dctr_param_count(self._h, C.byref(n)))
        out = {}
        for i in range(n.value):
            name = C.c_char_p()
            rank = C.c_int()
            dims = (C.c_int64 * 4)()
            capi.check(self._lib.dctr_param_info(self._h, i, C.byref(name), C.byref(rank), C.byref(dims)))
            out[name.value.decode()] = tuple(int(dims[k]) for k in range(rank.value))
        return out

    @property
    def param_shapes(self) -> Dict[str, Tuple[int, ...]]:
        return dict(self._shapes)

    def set_param(self, name: str, value) -> None:
        a = np.ascontiguousarray(np.asarray(value, dtype=np.float32))
        if tuple(a.shape) != self._shapes.get(name, None):
            raise errors.InvalidArgumentError("shape mismatch for %s: %s vs %s" % (name, a.shape, self._shapes.get(name)))
        capi.check(self._lib.dctr_param_set(self._h, name.encode(), capi.ptr(a), a.nbytes))

    def get_param(self, name: str) -> np.ndarray:
        a = np.empty(self._shapes[name], dtype=np.float32)
        capi.check(self._lib.dctr_param_get(self._h, name.encode(), capi.ptr(a), a.nbytes))
        return a

    def set_params(self, params: Dict[str, "np.ndarray"]) -> None:
        for k, v in params.items():
            self.set_param(k, v.detach().cpu().numpy() if hasattr(v, "detach") else v)

    def get_params(self) -> Dict[str, np.ndarray]:
        return {k: self.get_param(k) for k in self._shapes}

    def get_grad(self, name: str) -> np.ndarray:
        """Gradient of a dense variable as of the last backward pass (dctr_param_grad_get)."""
        a = np.empty(self._shapes[name], dtype=np.float32)
        capi.check(self._lib.dctr_param_grad_get(self._h, name.encode(), capi.ptr(a), a.nbytes))
        return a

    def afm_fwd(self, e, train: bool = False, want_att: bool = False, stream=None):
        """AFM.py:127-158 as an op on an afm handle: e [B, F*K] device tensor -> (y_emb [B, K], att [B, P] or None)."""
        import torch
        B, K, F = int(e.shape[0]), self.cfg.embedding_size, self.cfg.field_size
        y = torch.empty(B, K, device=e.device)
        att = torch.empty(B, F * (F - 1) // 2, device=e.device) if want_att else None
        st = stream if stream is not None else capi.current_stream()
        capi.check(self._lib.dctr_afm_fwd(self._h, capi.ptr(e), int(e.shape[1]), B, int(train), capi.ptr(y), K, capi.ptr(att), st))
        return y, att

    def afm_bwd(self, dy_emb, stream=None):
        """dL/d y_emb [B, K] -> dL/de [B, F*K]; the attention variables' gradients: get_grad(name)."""
        import torch
        B, K, F = int(dy_emb.shape[0]), self.cfg.embedding_size, self.cfg.field_size
        dE = torch.empty(B, F * K, device=dy_emb.device)
        st = stream if stream is not None else capi.current_stream()
        capi.check(self._lib.dctr_afm_bwd(self._h, capi.ptr(dy_emb), K, B, capi.ptr(dE), F * K, st))
        return dE

    def get_slot(self, name: str, which: int) -> np.ndarray:
        a = np.empty(self._shapes[name], dtype=np.float32)
        capi.check(self._lib.dctr_slot_get(self._h, name.encode(), which, capi.ptr(a), a.nbytes))
        return a

    def set_slot(self, name: str, which: int, value) -> None:
        a = np.ascontiguousarray(np.asarray(value, dtype=np.float32))
        capi.check(self._lib.dctr_slot_set(self._h, name.encode(), which, capi.ptr(a), a.nbytes))

    @property
    def global_step(self) -> int:
        s = C.c_int64()
        capi.check(self._lib.dctr_get_global_step(self._h, C.byref(s)))
        return s.value

    @global_step.setter
    def global_step(self, v: int) -> None:
        capi.check(self._lib.dctr_set_global_step(self._h, int(v)))

    def dropout_mask(self, site: int, shape, keep: float, step: Optional[int] = None) -> np.ndarray:
        """The 0/1 keep mask (uint8, `shape`) the engine applies at dropout site `site` (capi.SITE_*) in train step `step`
        (default: the next train_step).  Evaluated on the host by dctr_dropout_mask -- the engine's dropout is a pure function of
        (seed, step, site, element index), see include/deepctr_hip.h "dropout sites"."""
        m = np.empty(tuple(int(d) for d in shape), dtype=np.uint8)
        t = self.global_step + 1 if step is None else int(step)
        capi.check(self._lib.dctr_dropout_mask(int(self.cfg.seed), t, int(site), int(m.size), float(keep), capi.ptr(m)))
        return m

    # -- ops ---------------------------------------------------------------------------------
    def _set_dense(self, dense) -> None:
        if dense is not None:
            capi.check(self._lib.dctr_set_dense_input(self._h, capi.ptr(dense)))

    def train_step(self, ids, vals, labels, want_loss: bool = True, stream=None, dense=None) -> Optional[float]:
        """ids int32 [B,F], vals f32 [B,F], labels f32 [B] (dense f32 [B,dense_size] for the canned models): device tensors."""
        self._set_dense(dense)
        B = int(labels.shape[0])
        loss = C.c_float()
        st = stream if stream is not None else capi.current_stream()
        capi.check(self._lib.dctr_train_step(self._h, capi.ptr(ids), capi.ptr(vals), capi.ptr(labels), B,
                                             C.byref(loss) if want_loss else None, st))
        return loss.value if want_loss else None

    def train_step_csr(self, offsets, ids, weights, y, z=None, want_loss: bool = True, stream=None) -> Optional[float]:
        """CSR models (din / esmm): offsets int32 [B*S+1], ids int32 [nnz], weights f32 [nnz] or None, labels y (and z for
        esmm) f32 [B]; all device tensors."""
        B = int(y.shape[0])
        loss = C.c_float()
        st = stream if stream is not None else capi.current_stream()
        capi.check(self._lib.dctr_train_step_csr(self._h, capi.ptr(offsets), capi.ptr(ids), capi.ptr(weights), int(ids.shape[0]),
                                                 capi.ptr(y), capi.ptr(z), B, C.byref(loss) if want_loss else None, st))
        return loss.value if want_loss else None

    def eval_batch_csr(self, offsets, ids, weights, y, z=None, stream=None) -> None:
        st = stream if stream is not None else capi.current_stream()
        capi.check(self._lib.dctr_eval_batch_csr(self._h, capi.ptr(offsets), capi.ptr(ids), capi.ptr(weights), int(ids.shape[0]),
                                                 capi.ptr(y), capi.ptr(z), int(y.shape[0]), st))

    def eval_auc_extra(self, which: int, stream=None) -> float:
        """esmm: which = 1 -> CVR_AUC (z, pcvr), 2 -> CTCVR_AUC (z, pctcvr); 0 -> the first output's AUC."""
        a = C.c_float()
        st = stream if stream is not None else capi.current_stream()
        capi.check(self._lib.dctr_eval_auc_extra(self._h, int(which), C.byref(a), st))
        return a.value

    def predict_csr(self, offsets, ids, weights, B: int, out0=None, out1=None, out2=None, stream=None):
        """din: out0 = prob, out1 = logit; esmm: out0 = pctr, out1 = pcvr, out2 = pctcvr (DeepCvrMTL.py:212)."""
        st = stream if stream is not None else capi.current_stream()
        capi.check(self._lib.dctr_predict_csr(self._h, capi.ptr(offsets), capi.ptr(ids), capi.ptr(weights), int(ids.shape[0]), int(B),
                                              capi.ptr(out0), capi.ptr(out1), capi.ptr(out2), st))
        return out0

    def prefetch_ids(self, ids_next) -> None:
        """Announces the next training batch's ids (a view of an input slot, unchanged until that train_step): grouped during the
        tail of the step in flight.  A scheduling hint only."""
        capi.check(self._lib.dctr_prefetch_ids(self._h, capi.ptr(ids_next), int(ids_next.shape[0])))

    def sync_tables(self, stream=None) -> None:
        """Advances every lagging table row to global_step (table_sweep_period > 1; a no-op otherwise)."""
        capi.check(self._lib.dctr_tables_sync(self._h, stream if stream is not None else capi.current_stream()))

    def prefetch_cancel(self) -> None:
        """Drops a pending prefetch_ids hint (the announced batch will not be trained)."""
        capi.check(self._lib.dctr_prefetch_cancel(self._h))

    def input_slot_rewrite(self, slot: int) -> None:
        """Call before refilling input slot `slot`: a grouping prefetched from its old contents is dropped (thread-safe)."""
        capi.check(self._lib.dctr_input_slot_rewrite(self._h, int(slot)))

    # ---- the H2D leg of the input pipeline inside the library (feeder.py; csrc/engine.hip dctr_input_slot_fill ...)
    def input_slot_fill(self, slot: int, h_ids: int, h_vals: int, h_labels: int, B: int) -> None:
        """B rows from HOST buffers (raw addresses; pinned for overlap) into input slot `slot` on the engine's copy stream.  Input thread."""
        capi.check(self._lib.dctr_input_slot_fill(self._h, int(slot), h_ids, h_vals, h_labels, int(B)))

    def input_slot_acquire(self, slot: int, stream=None) -> None:
        """`stream` (default: torch's current) waits on the device for the slot's last fill."""
        capi.check(self._lib.dctr_input_slot_acquire(self._h, int(slot), stream if stream is not None else capi.current_stream()))

    def input_slot_release(self, slot: int, stream=None) -> None:
        """Records "slot consumed" behind what has been enqueued on `stream`."""
        capi.check(self._lib.dctr_input_slot_release(self._h, int(slot), stream if stream is not None else capi.current_stream()))

    def input_slot_wait_released(self, slot: int) -> None:
        """Blocks the calling thread until the slot's last release has been reached on the device.  Input thread."""
        capi.check(self._lib.dctr_input_slot_wait_released(self._h, int(slot)))

    def input_slot_ready(self, slot: int) -> bool:
        import ctypes
        r = ctypes.c_int(1)
        capi.check(self._lib.dctr_input_slot_ready(self._h, int(slot), ctypes.byref(r)))
        return bool(r.value)

    def predict(self, ids, vals, out_prob=None, out_logit=None, stream=None, dense=None):
        self._set_dense(dense)
        B = int(ids.shape[0])
        st = stream if stream is not None else capi.current_stream()
        capi.check(self._lib.dctr_predict(self._h, capi.ptr(ids), capi.ptr(vals), B, capi.ptr(out_prob),
                                          capi.ptr(out_logit), st))
        return out_prob

    def main_stream(self):
        """torch view (ExternalStream) of the engine-owned stream for step calls (dctr_main_stream)."""
        import torch
        p = C.c_void_p()
        capi.check(self._lib.dctr_main_stream(self._h, C.byref(p)))
        return torch.cuda.ExternalStream(p.value)

    def input_slot(self, slot: int):
        """Zero-copy torch views (ids i32 [max_batch,F], vals f32 [max_batch,F], labels f32 [max_batch]) of engine-owned
        input staging set `slot`; batches written there are consumed by train_step/predict without a staging copy."""
        import torch
        pi, pv, pl = C.c_void_p(), C.c_void_p(), C.c_void_p()
        capi.check(self._lib.dctr_input_slot(self._h, slot, C.byref(pi), C.byref(pv), C.byref(pl)))
        dev = torch.device("cuda", torch.cuda.current_device())
        MB, F = self.cfg.max_batch, self.cfg.field_size

        def view(ptr, shape, typestr):
            class _A:
                pass
            a = _A()
            a.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr.value, False), "version": 2}
            return torch.as_tensor(a, device=dev)
        return view(pi, (MB, F), "<i4"), view(pv, (MB, F), "<f4"), view(pl, (MB,), "<f4")

    def eval_reset(self, stream=None) -> None:
        capi.check(self._lib.dctr_eval_reset(self._h, stream if stream is not None else capi.current_stream()))

    def eval_batch(self, ids, vals, labels, stream=None, dense=None) -> None:
        self._set_dense(dense)
        st = stream if stream is not None else capi.current_stream()
        capi.check(self._lib.dctr_eval_batch(self._h, capi.ptr(ids), capi.ptr(vals), capi.ptr(labels), int(labels.shape[0]), st))

    def eval_result(self, stream=None):
        """-> (auc, loss, n_examples) accumulated since eval_reset (tf.metrics.auc semantics, DeepFM.py:193-195)."""
        auc, loss, n = C.c_float(), C.c_float(), C.c_int64()
        st = stream if stream is not None else capi.current_stream()
        capi.check(self._lib.dctr_eval_result(self._h, C.byref(auc), C.byref(loss), C.byref(n), st))
        return auc.value, loss.value, n.value

    def check_ids(self, stream=None) -> None:
        st = stream if stream is not None else capi.current_stream()
        capi.check(self._lib.dctr_check_ids(self._h, st))

    def time_stage(self, stage: str, iters: int = 50, stream=None) -> float:
        """Average milliseconds per execution of one stage of the step (hipEvents around a graph of `iters` launches)."""
        ms = C.c_float()
        st = stream if stream is not None else capi.current_stream()
        capi.check(self._lib.dctr_time_kernel(self._h, stage.encode(), iters, C.byref(ms), st))
        return ms.value

    def step_timer(self, enable):
        """Arms (True / 1: a bracket around the first MLP layer's forward GEMM; 2: every forward layer's own dispatch events) /
        reads (False -> (avg_ms, count) over all timed launches) the in-step timer (dctr_step_timer)."""
        ms, n = C.c_float(), C.c_int()
        capi.check(self._lib.dctr_step_timer(self._h, int(enable), C.byref(ms), C.byref(n)))
        return None if enable else (ms.value, n.value)

    def step_timer_layer(self, layer: int):
        """(avg_ms, count) of one MLP layer's timed forward launches, after step_timer(False)."""
        ms, n = C.c_float(), C.c_int()
        capi.check(self._lib.dctr_step_timer_layer(self._h, int(layer), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    @staticmethod
    def measure_copy_bandwidth(nbytes: int = 1 << 30, iters: int = 20, stream=None) -> float:
        """GB/s (read + write) of a streaming float4 copy: the measured HBM roofline of this box."""
        g = C.c_float()
        st = stream if stream is not None else capi.current_stream()
        capi.check(capi.lib().dctr_measure_copy_bw(int(nbytes), int(iters), C.byref(g), st))
        return g.value

    def param_tensor(self, name: str):
        """Zero-copy torch view of a parameter in device memory (dctr_param_device_view): device-side initialisation of tables that
        are too large to stage through the host.  A table kept as row records (include/deepctr_hip.h) comes back as a STRIDED view
        (`.is_contiguous()` False: index it by rows, do not `.view(-1)` it)."""
        import torch
        p = C.c_void_p()
        ld = C.c_int64()
        capi.check(self._lib.dctr_param_device_view(self._h, name.encode(), C.byref(p), C.byref(ld)))
        shape = tuple(int(d) for d in self.param_shapes[name])
        iface = {"shape": shape, "typestr": "<f4", "data": (p.value, False), "version": 2}
        inner = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        if len(shape) >= 1 and ld.value != inner:
            strides, acc = [], 4
            for d in reversed(shape[1:]):
                strides.insert(0, acc)
                acc *= d
            iface["strides"] = tuple([4 * ld.value] + strides)

        class _A:
            pass
        a = _A()
        a.__cuda_array_interface__ = iface
        return torch.as_tensor(a, device=torch.device("cuda", torch.cuda.current_device()))

    def debug_tensor(self, name: str):
        """Device view (torch) of a named intermediate of the last forward."""
        import torch
        p = C.c_void_p()
        n = C.c_int64()
        ld = C.c_int()
        capi.check(self._lib.dctr_debug_tensor(self._h, name.encode(), C.byref(p), C.byref(n), C.byref(ld)))
        host = np.empty(n.value, dtype=np.float32)
        capi.check(self._lib.dctr_memcpy_d2h(capi.ptr(host), p, host.nbytes, None))
        if ld.value > 1:
            host = host.reshape(-1, ld.value)
        return torch.from_numpy(host)
