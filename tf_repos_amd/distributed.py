"""Row-sharded multi-GPU training: one process per GPU, tables sharded by row (owner = id % world,
local row = id // world), dense parameters replicated, synchronous steps.

This replaces the reference's only parallelism -- TF's asynchronous parameter server configured by set_dist_env
(DeepFM.py:237-282, run_dist.sh) -- with the MI355X-native equivalent.  Per step, over xGMI:

  routing  (depends only on the batch's ids; runs ONE STEP AHEAD on a side stream + its own communicator):
           de-duplicate ids -> bucket by owner -> all-gather of the split sizes (the only host sync) -> all-to-all of
           the distinct local rows -> owner groups the requested rows for the later gradient segment-sum
  rows     owner packs [row | linear weight] records -> ONE all-to-all back -> forward + backward
  grads    per-distinct-id gradients packed the same way -> ONE all-to-all -> owner segment-sum + optimizer on its shard
  dense    flat gradient arena (~2-3 MB) -> all-reduce on a third stream/communicator, beside the gradient exchange

Every GPU talks to its 7 peers at once in the all-to-alls, so all 7 links carry traffic; an id crosses the fabric once
per direction however often the batch repeats it.  Rows are always fetched AFTER the previous step's update, so the
step is synchronous SGD (N ranks == 1 rank on the same global batch, tests/test_distributed.py).

torch.distributed ("nccl" == RCCL on ROCm) is the transport only; every arithmetic op is a libdeepctr_hip.so call.
`Comm` can also stage through host memory over gloo so that the same code runs in 2-process tests on one GPU.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from . import capi
from .engine import Engine, EngineConfig


class Comm:
    """Variable-split all-to-all / all-reduce on device tensors.  backend 'nccl': tensors go to RCCL as they are;
    'gloo': staged through host memory (tests)."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.staged = dist.get_backend(group) != "nccl"

    def all_counts(self, counts) -> np.ndarray:
        """counts: this rank's `world` send split sizes (device/host tensor or list) -> int matrix [world, world] whose row s
        is rank s's send counts (so column r = what rank r receives).  One host synchronisation."""
        W = self.world
        if not torch.is_tensor(counts):
            counts = torch.tensor(list(counts), dtype=torch.int32)
        counts = counts[:W].to(torch.int32)
        if self.staged:
            parts = [torch.empty(W, dtype=torch.int32) for _ in range(W)]
            dist.all_gather(parts, counts.cpu().contiguous(), group=self.group)
            return torch.stack(parts).numpy().astype(np.int64)
        out = torch.empty(W * W, dtype=torch.int32, device=counts.device)
        dist.all_gather_into_tensor(out, counts.contiguous(), group=self.group)
        return out.cpu().numpy().reshape(W, W).astype(np.int64)

    def all_to_all(self, send: torch.Tensor, send_counts: Sequence[int], recv_counts: Sequence[int],
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """send: [sum(send_counts), ...] grouped by destination -> [sum(recv_counts), ...] grouped by source
        (written into `out[:n]` when given)."""
        n = int(sum(recv_counts))
        shape = (n,) + tuple(send.shape[1:])
        if out is None:
            out = torch.empty(shape, dtype=send.dtype, device=send.device)
        else:
            out = out[:n]
        if self.staged:
            h = torch.empty(shape, dtype=send.dtype)
            dist.all_to_all_single(h, send.cpu().contiguous(), list(recv_counts), list(send_counts), group=self.group)
            out.copy_(h)
        else:
            dist.all_to_all_single(out, send, list(recv_counts), list(send_counts), group=self.group)
        return out

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        if self.staged:
            h = t.cpu()
            dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, group=self.group)
        return t


@dataclass
class Route:
    """Where one batch's distinct ids live: split sizes of the exchanges and the local rows this rank must serve."""
    send_counts: List[int]
    recv_counts: List[int]
    recv_rows: torch.Tensor                     # [n_recv] local rows requested from this rank, grouped by requester
    parity: int = 0
    ids_key: int = 0
    ids_ref: object = None      # the prefetched ids tensor itself: a freed tensor's address can be handed to a new one
    event: Optional[object] = None              # recorded on the stream that computed the route

    @property
    def n_send(self) -> int:
        return int(sum(self.send_counts))

    @property
    def n_recv(self) -> int:
        return int(sum(self.recv_counts))


class ShardExchange:
    """The exchange protocol, independent of what computes: route distinct ids to their owners, bring rows back, send
    gradients the reverse way."""

    def __init__(self, comm: Comm):
        self.comm = comm

    def route(self, send_rows: torch.Tensor, counts, out: Optional[torch.Tensor] = None) -> Route:
        """send_rows: local rows grouped by owner; counts: the per-owner split sizes (device tensor or list)."""
        c = self.comm
        m = c.all_counts(counts)
        send_counts = [int(x) for x in m[c.rank]]
        recv_counts = [int(x) for x in m[:, c.rank]]
        recv_rows = c.all_to_all(send_rows[:sum(send_counts)], send_counts, recv_counts, out=out)
        return Route(send_counts, recv_counts, recv_rows)

    def fetch(self, r: Route, answers: torch.Tensor, out: Optional[torch.Tensor] = None, comm: Optional[Comm] = None) -> torch.Tensor:
        """answers [n_recv, ...] (one record per requested row, same order as r.recv_rows) -> [n_send, ...] in send order."""
        return (comm or self.comm).all_to_all(answers[:r.n_recv], r.recv_counts, r.send_counts, out=out)

    def return_grads(self, r: Route, grads: torch.Tensor, out: Optional[torch.Tensor] = None, comm: Optional[Comm] = None) -> torch.Tensor:
        """grads [n_send, ...] in send order -> [n_recv, ...] aligned with r.recv_rows on the owner."""
        return (comm or self.comm).all_to_all(grads[:r.n_send], r.send_counts, r.recv_counts, out=out)


class ShardedTrainer:
    """One rank of the row-sharded trainer.  `workload` carries the reference flags (model, field_size, feature_size,
    embedding_size, batch (per rank), deep_layers, dropout, l2_reg, learning_rate, optimizer, cross_layers).

    driver="native" (default): the step is enqueued by the C++ driver (csrc/dist.hip) -- over RCCL communicators of its own
    when the process group's backend is nccl, else over host-staged callbacks into this module (gloo tests).
    driver="python": the same protocol orchestrated from here through torch.distributed (readable reference, slower host
    side).  overlap=True routes the next batch / all-reduces the dense gradients on side streams with their own
    communicators; every rank must construct the trainer at the same point (communicator creation is collective)."""

    def __init__(self, workload: Dict, rank: int, world: int, device: torch.device, table_mode: str = "dense_exact",
                 seed: int = 1, group=None, init_scale: float = 0.01, params: Optional[Dict[str, np.ndarray]] = None,
                 overlap: Optional[bool] = None, driver: Optional[str] = None, init_tables: bool = True):
        self.w = dict(workload)
        self.init_tables = init_tables        # False: the caller fills the table shards on the device (a 1e8-row table never exists on the host)
        self.rank, self.world, self.dev = rank, world, device
        self.comm = Comm(group)
        assert self.comm.world == world and self.comm.rank == rank
        if overlap is None:
            overlap = os.environ.get("DCTR_SHARD_OVERLAP", "1") != "0"
        self.overlap = bool(overlap)
        self.driver = driver or os.environ.get("DCTR_SHARD_DRIVER", "native")
        if self.driver not in ("native", "python"):
            raise ValueError("driver must be 'native' or 'python', got %r" % self.driver)
        self._dist = None
        if self.driver == "native":
            self.comm_route = self.comm_dense = self.comm
            self.s_route = self.s_dense = None
        elif self.overlap:
            ranks = dist.get_process_group_ranks(group) if group is not None else None
            self.comm_route = Comm(dist.new_group(ranks=ranks))
            self.comm_dense = Comm(dist.new_group(ranks=ranks))
            self.s_route = torch.cuda.Stream(device=device)
            self.s_dense = torch.cuda.Stream(device=device)
        else:
            self.comm_route = self.comm_dense = self.comm
            self.s_route = self.s_dense = None
        self.x = ShardExchange(self.comm)
        self.x_route = ShardExchange(self.comm_route)
        w = self.w
        self.F, self.K, self.V, self.B = w["field_size"], w["embedding_size"], w["feature_size"], w["batch"]
        self.P = self.K + 4                      # floats per packed row record
        self.eng = Engine(EngineConfig(model=w["model"], field_size=self.F, feature_size=self.V, embedding_size=self.K,
                                       deep_layers=w["deep_layers"], dropout=w["dropout"], cross_layers=w.get("cross_layers", 3),
                                       l2_reg=w["l2_reg"], learning_rate=w["learning_rate"], optimizer=w["optimizer"],
                                       table_mode=table_mode, max_batch=self.B, seed=seed, shard_rank=rank, shard_world=world,
                                       use_graph=False, batch_norm=bool(w.get("batch_norm", False)),
                                       batch_norm_decay=float(w.get("batch_norm_decay", 0.9)),
                                       # (the Python orchestration is the readable reference of the protocol: it keeps the classic
                                       #  sweep of every owned row every step; the native driver's owner side lags, csrc/lag.h)
                                       table_sweep_period=(int(w.get("table_sweep_period", 0)) if self.driver == "native" else 1)))
        self._lib = capi.lib()
        self._h = self.eng._h
        self.init_params(params, init_scale, seed)
        if self.driver == "native":
            self._create_native()
            return
        self._install_stat_sync()
        cap = self.B * self.F                    # distinct ids this rank can request
        cap_owner = cap * world                  # rows this rank can be asked for
        i32 = dict(dtype=torch.int32, device=device)
        f32 = dict(dtype=torch.float32, device=device)
        # two routing states (parity of the step): the next batch is routed while the current one trains
        self._g = [C.c_void_p(), C.c_void_p()]   # requester-side grouping over the GLOBAL id space
        for g in self._g:
            capi.check(self._lib.dctr_group_create(self.V, cap, self.K, C.byref(g)))
        self.send_rows = [torch.empty(cap, **i32) for _ in range(2)]
        self.upos = [torch.empty(cap, **i32) for _ in range(2)]
        self.idx = [torch.empty(cap, **i32) for _ in range(2)]
        self.counts = [torch.zeros(2 * world, **i32) for _ in range(2)]
        self.recv_rows = [torch.empty(cap_owner, **i32) for _ in range(2)]
        self._done = [None, None]                # event: the last step that used this parity has finished
        self._parity = 0
        self._pref: Optional[Route] = None
        self.rows_out = torch.empty(cap_owner, self.P, **f32)
        self.rows_back = torch.empty(cap, self.P, **f32)
        self.send_grads = torch.empty(cap, self.P, **f32)
        self.recv_grads = torch.empty(cap_owner, self.P, **f32)

    def _install_stat_sync(self) -> None:
        """Python driver: batch_norm's column sums through this process group (the native driver installs its transport's)."""
        self._stat_exc = None

        def cb(ctx, ch, d_buf, n, stream):
            try:
                # the reduction runs on torch's current stream: that must be the stream the engine enqueues the step on
                if int(stream or 0) != int(torch.cuda.current_stream().cuda_stream):
                    raise RuntimeError("batch_norm statistics: the engine's stream is not torch's current stream")
                self.comm.all_reduce_sum(_as_tensor(d_buf, int(n), self.dev))
                return 0
            except BaseException as e:      # never let an exception unwind through the C frames
                self._stat_exc = e
                return 5
        self._stat_cb = capi.ALL_REDUCE_F32_FN(cb)
        capi.check(self._lib.dctr_set_stat_sync(self._h, self._stat_cb, None, self.world))

    def _create_native(self) -> None:
        L = self._lib
        self._dist = C.c_void_p()
        os.environ["DCTR_SHARD_OVERLAP"] = "1" if self.overlap else "0"        # read by dctr_dist_create
        if not self.comm.staged:
            rccl_path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so").encode()
            blob = [None]
            if self.rank == 0:
                ids = C.create_string_buffer(3 * capi.RCCL_ID_BYTES)
                for ch in range(3):
                    capi.check(L.dctr_rccl_unique_id(rccl_path, C.cast(C.byref(ids, ch * capi.RCCL_ID_BYTES), C.c_char_p)))
                blob[0] = ids.raw
            dist.broadcast_object_list(blob, src=dist.get_global_rank(self.comm.group, 0) if self.comm.group is not None else 0,
                                       group=self.comm.group)
            capi.check(L.dctr_dist_create_rccl(self._h, self.rank, self.world, blob[0], rccl_path, C.byref(self._dist)))
            self._transport = None
        else:
            self._transport = HostStagedTransport(self.comm)
            capi.check(L.dctr_dist_create(self._h, self.rank, self.world, C.byref(self._transport.table), C.byref(self._dist)))

    # -- parameters: every rank draws the same full tensors and keeps its rows -----------------------------------
    def init_params(self, params: Optional[Dict[str, np.ndarray]], scale: float, seed: int) -> None:
        rng = np.random.default_rng(seed)
        full_rows = self.V
        for name, shp in self.eng.param_shapes.items():
            if name in ("emb", "linear"):
                if not self.init_tables and params is None:
                    continue
                full = (full_rows,) + tuple(shp[1:])
                a = params[name] if params is not None else rng.normal(0, scale, size=full).astype(np.float32)
                self.eng.set_param(name, np.ascontiguousarray(np.asarray(a)[self.rank::self.world]))
            else:
                a = params[name] if params is not None else rng.normal(0, scale, size=shp).astype(np.float32)
                self.eng.set_param(name, np.asarray(a))

    def gather_full_params(self) -> Dict[str, np.ndarray]:
        """All-gathers the table shards back into full [V, ...] arrays (tests / checkpoints)."""
        out = {}
        for name in self.eng.param_shapes:
            a = self.eng.get_param(name)
            if name in ("emb", "linear"):
                parts = [None] * self.world
                dist.all_gather_object(parts, a, group=self.comm.group)
                full = np.empty((self.V,) + a.shape[1:], dtype=np.float32)
                for r, p in enumerate(parts):
                    full[r::self.world] = p
                out[name] = full
            else:
                out[name] = a
        return out

    # -- routing ----------------------------------------------------------------------------------------------------------
    def _route(self, ids: torch.Tensor) -> Route:
        """Everything that depends only on the ids, on the CURRENT stream: requester-side de-duplication and bucketing,
        split sizes, the local rows to their owners, owner-side grouping of the requested rows."""
        L, st = self._lib, capi.current_stream()
        par = self._parity
        self._parity ^= 1
        if self._done[par] is not None:
            torch.cuda.current_stream().wait_event(self._done[par])      # this parity's buffers are free again
        B = int(ids.shape[0])
        g = self._g[par]
        capi.check(L.dctr_group_ids(g, capi.ptr(ids), B, self.F, st))
        capi.check(L.dctr_route_unique(g, self.world, capi.ptr(self.send_rows[par]), capi.ptr(self.upos[par]),
                                       capi.ptr(self.counts[par]), st))
        r = self.x_route.route(self.send_rows[par], self.counts[par], out=self.recv_rows[par])      # the one host sync
        capi.check(L.dctr_entry_index(g, capi.ptr(ids), B * self.F, capi.ptr(self.upos[par]), capi.ptr(self.idx[par]), st))
        capi.check(L.dctr_table_group_rows(self._h, par, capi.ptr(self.recv_rows[par]), r.n_recv, st))
        r.parity, r.ids_key, r.ids_ref = par, ids.data_ptr(), ids
        return r

    def prefetch(self, next_ids: torch.Tensor) -> None:
        """Routes `next_ids` (the ids of the batch the NEXT train_step/predict call will get) on the routing stream while
        the step just enqueued runs.  `next_ids` must have been produced before the current step was enqueued."""
        if self.s_route is None:
            return
        self.s_route.wait_event(self._ev_start)
        with torch.cuda.stream(self.s_route):
            r = self._route(next_ids)
            r.event = torch.cuda.Event()
            r.event.record(self.s_route)
        self._pref = r

    def _take_route(self, ids: torch.Tensor) -> Route:
        self._ev_start = torch.cuda.Event()
        self._ev_start.record(torch.cuda.current_stream())
        r, self._pref = self._pref, None
        # a prefetched route belongs to THIS tensor (same storage, same view), not merely to its address: torch's caching allocator
        # reuses freed addresses, and a new ids tensor at the old address with other contents must be routed afresh
        if r is not None and (r.ids_ref is ids or (r.ids_ref is not None and r.ids_key == ids.data_ptr() and r.ids_ref.shape == ids.shape
                                                   and r.ids_ref.untyped_storage().data_ptr() == ids.untyped_storage().data_ptr()
                                                   and r.ids_ref._version == ids._version)):
            torch.cuda.current_stream().wait_event(r.event)
            return r
        if r is not None:                       # a prefetch for some other batch: wait for it, then drop it
            torch.cuda.current_stream().wait_event(r.event)
        return self._route(ids)

    # -- one synchronous step -------------------------------------------------------------------------------------------
    def _forward_backward(self, r: Route, vals, labels, B: int, train: bool) -> None:
        L, st = self._lib, capi.current_stream()
        par = r.parity
        capi.check(L.dctr_table_gather_packed(self._h, capi.ptr(self.recv_rows[par]), r.n_recv, capi.ptr(self.rows_out), st))
        rows = self.x.fetch(r, self.rows_out, out=self.rows_back)
        rc = L.dctr_sharded_forward_backward(self._h, capi.ptr(rows), r.n_send, capi.ptr(self.idx[par]), capi.ptr(vals),
                                             capi.ptr(labels), B, B * self.world, int(train), st)
        exc, self._stat_exc = getattr(self, "_stat_exc", None), None
        if exc is not None:           # the batch_norm statistics callback failed: surface ITS error, not the engine's generic status
            raise exc
        capi.check(rc)

    def _finish(self, r: Route) -> None:
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._done[r.parity] = ev

    def train_step(self, ids, vals, labels, want_loss: bool = False, next_ids=None) -> Optional[float]:
        """One synchronous step on this rank's examples.  next_ids: the ids tensor the NEXT call will be given (already written;
        left untouched until then) -- routed while this step runs."""
        L = self._lib
        if self._dist is not None:
            loss = C.c_float()
            capi.check(L.dctr_dist_train_step(self._dist, capi.ptr(ids), capi.ptr(vals), capi.ptr(labels), int(ids.shape[0]),
                                              capi.ptr(next_ids), int(next_ids.shape[0]) if next_ids is not None else 0,
                                              C.byref(loss) if want_loss else None, capi.current_stream()))
            if self._transport is not None:
                self._transport.reraise()
            return loss.value if want_loss else None
        M = torch.cuda.current_stream()
        st = capi.current_stream()
        B = int(ids.shape[0])
        r = self._take_route(ids)
        self._forward_backward(r, vals, labels, B, True)
        # dense side (beside the gradient exchange): flat gradient arena -> all-reduce (sum; the logit gradient already
        # carries 1/global_batch) -> optimizer
        if self.s_dense is not None:
            self.s_dense.wait_stream(M)
            with torch.cuda.stream(self.s_dense):
                self._dense_update()
        # sparse side: per-distinct-id gradients in send order -> owners -> segment-sum + table optimizer
        capi.check(L.dctr_sharded_pack_row_grads(self._h, self._g[r.parity], B, capi.ptr(self.upos[r.parity]),
                                                 capi.ptr(self.send_grads), st))
        grads = self.x.return_grads(r, self.send_grads, out=self.recv_grads)
        capi.check(L.dctr_table_apply_packed(self._h, r.parity, r.n_recv, capi.ptr(grads), st))
        if self.s_dense is not None:
            M.wait_stream(self.s_dense)
        else:
            self._dense_update()
        self._finish(r)
        if next_ids is not None:
            self.prefetch(next_ids)
        if not want_loss:
            return None
        sc = (C.c_float * 4)()
        capi.check(L.dctr_read_scalars(self._h, C.byref(sc), st))
        t = torch.tensor([sc[0], sc[1], sc[2]], dtype=torch.float64, device=self.dev)
        self.comm.all_reduce_sum(t)
        xent, sq_emb, sq_lin = (float(v) for v in t.cpu())
        return xent / (B * self.world) + self.w["l2_reg"] * 0.5 * (sq_emb + sq_lin + float(sc[3]))

    def _dense_update(self) -> None:
        L, st = self._lib, capi.current_stream()
        flat, nflat = C.c_void_p(), C.c_int64()
        capi.check(L.dctr_dense_grads(self._h, C.byref(flat), C.byref(nflat), st))
        self.comm_dense.all_reduce_sum(_as_tensor(flat.value, nflat.value, self.dev))
        capi.check(L.dctr_dense_apply(self._h, st))

    def predict(self, ids, vals) -> torch.Tensor:
        B = int(ids.shape[0])
        if self._dist is not None:
            out = torch.empty(B, dtype=torch.float32, device=self.dev)
            capi.check(self._lib.dctr_dist_predict(self._dist, capi.ptr(ids), capi.ptr(vals), B, capi.ptr(out), capi.current_stream()))
            if self._transport is not None:
                self._transport.reraise()
            return out
        r = self._take_route(ids)
        self._forward_backward(r, vals, None, B, False)
        self._finish(r)
        p = C.c_void_p()
        capi.check(self._lib.dctr_last_outputs(self._h, C.byref(p), None))
        return _as_tensor(p.value, B, self.dev).clone()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def close(self):
        if self._dist is not None and self._dist.value:
            self._lib.dctr_dist_destroy(self._dist)
            self._dist = None
        for g in getattr(self, "_g", []):
            if g is not None and g.value:
                self._lib.dctr_group_destroy(g)
        self._g = []
        self.eng.close()


class HostStagedTransport:
    """dctr_transport callbacks that stage through host memory over a (gloo) process group: lets the native driver run with
    several ranks sharing one GPU (tests).  Every callback is synchronous: stream sync, D2H, collective, H2D, stream sync."""

    def __init__(self, comm: Comm):
        self.comm = comm
        self._lib = capi.lib()
        self._exc = None
        # channel 1 (routing) is driven by the native driver's worker thread, concurrently with the main thread's channels 0 / 2:
        # it needs a process group of its own so that the two threads' collectives cannot interleave differently across ranks
        ranks = dist.get_process_group_ranks(comm.group) if comm.group is not None else None
        self._groups = {0: comm.group, 1: dist.new_group(ranks=ranks), 2: comm.group}
        self._cbs = (capi.ALL_GATHER_I32_FN(self._all_gather_i32), capi.ALL_TO_ALL_FN(self._all_to_all),
                     capi.ALL_REDUCE_F32_FN(self._all_reduce_f32))            # keep the thunks alive
        self.table = capi.Transport(None, *self._cbs)

    def reraise(self):
        e, self._exc = self._exc, None
        if e is not None:
            raise e

    def _d2h(self, dptr, nbytes, dtype, stream) -> np.ndarray:
        host = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        if nbytes:
            capi.check(self._lib.dctr_memcpy_d2h(capi.ptr(host), C.c_void_p(dptr), nbytes, C.c_void_p(stream)))
        capi.check(self._lib.dctr_stream_sync(C.c_void_p(stream)))
        return host

    def _h2d(self, dptr, host: np.ndarray, stream) -> None:
        if host.nbytes:
            capi.check(self._lib.dctr_memcpy_h2d(C.c_void_p(dptr), capi.ptr(host), host.nbytes, C.c_void_p(stream)))
        capi.check(self._lib.dctr_stream_sync(C.c_void_p(stream)))

    def _guard(self, fn):
        try:
            fn()
            return 0
        except BaseException as e:          # never let an exception unwind through the C frames
            self._exc = e
            return 5

    def _all_gather_i32(self, ctx, ch, d_send, n, d_recv, stream):
        def run():
            mine = torch.from_numpy(self._d2h(d_send, 4 * n, np.int32, stream))
            parts = [torch.empty(n, dtype=torch.int32) for _ in range(self.comm.world)]
            dist.all_gather(parts, mine, group=self._groups[ch])
            self._h2d(d_recv, torch.cat(parts).numpy(), stream)
        return self._guard(run)

    def _all_to_all(self, ctx, ch, d_send, scnt, d_recv, rcnt, rec, stream):
        def run():
            W = self.comm.world
            sc = [int(scnt[p]) for p in range(W)]
            rc = [int(rcnt[p]) for p in range(W)]
            send = torch.from_numpy(self._d2h(d_send, sum(sc) * rec, np.uint8, stream)).reshape(sum(sc), rec)
            recv = torch.empty(sum(rc), rec, dtype=torch.uint8)
            dist.all_to_all_single(recv, send, rc, sc, group=self._groups[ch])
            self._h2d(d_recv, recv.numpy(), stream)
        return self._guard(run)

    def _all_reduce_f32(self, ctx, ch, d_buf, n, stream):
        def run():
            h = torch.from_numpy(self._d2h(d_buf, 4 * n, np.float32, stream))
            dist.all_reduce(h, group=self._groups[ch])
            self._h2d(d_buf, h.numpy(), stream)
        return self._guard(run)


def _as_tensor(ptr: int, n: int, dev: torch.device) -> torch.Tensor:
    """Zero-copy torch view of `n` floats of engine-owned device memory (for RCCL calls)."""
    class _Arr:
        pass
    a = _Arr()
    a.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}
    return torch.as_tensor(a, device=dev)
