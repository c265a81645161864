"""Row-sharded multi-GPU training: one process per GPU, tables sharded by row (owner = id % world,
local row = id // world), dense parameters replicated, synchronous steps.

This replaces the reference's only parallelism -- TF's asynchronous parameter server configured by set_dist_env
(DeepFM.py:237-282, run_dist.sh) -- with the MI355X-native equivalent: per step three all-to-alls over xGMI
(distinct local rows out, rows back, row gradients back; every GPU talks to its 7 peers at once so all 7 links carry
traffic) plus one all-reduce of the ~2-3 MB dense gradient arena.  The batch's ids are de-duplicated before they are
routed, so an id crosses the fabric once per direction however often the batch repeats it.

torch.distributed ("nccl" == RCCL on ROCm) is the transport only; every arithmetic op is a libdeepctr_hip.so call.
`Comm` can also stage through host memory over gloo so that the same code runs in 2-process tests on one GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import capi
from .engine import Engine, EngineConfig


class Comm:
    """Variable-split all-to-all / all-reduce on device tensors.  backend 'nccl': tensors go to RCCL as they are;
    'gloo': staged through pinned host memory (tests)."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.staged = dist.get_backend(group) != "nccl"

    def exchange_counts(self, send_counts: Sequence[int]) -> List[int]:
        s = torch.tensor(list(send_counts), dtype=torch.int64)
        r = torch.empty_like(s)
        if self.staged:
            dist.all_to_all_single(r, s, group=self.group)
        else:
            dev = torch.device("cuda", torch.cuda.current_device())
            sd, rd = s.to(dev), r.to(dev)
            dist.all_to_all_single(rd, sd, group=self.group)
            r = rd.cpu()
        return [int(x) for x in r]

    def all_to_all(self, send: torch.Tensor, send_counts: Sequence[int], recv_counts: Sequence[int]) -> torch.Tensor:
        """send: [sum(send_counts), ...] grouped by destination -> [sum(recv_counts), ...] grouped by source."""
        shape = (int(sum(recv_counts)),) + tuple(send.shape[1:])
        if self.staged:
            out = torch.empty(shape, dtype=send.dtype)
            dist.all_to_all_single(out, send.cpu().contiguous(), list(recv_counts), list(send_counts), group=self.group)
            return out.to(send.device)
        out = torch.empty(shape, dtype=send.dtype, device=send.device)
        dist.all_to_all_single(out, send.contiguous(), list(recv_counts), list(send_counts), group=self.group)
        return out

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        if self.staged:
            h = t.cpu()
            dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, group=self.group)
        return t


class ShardExchange:
    """The exchange protocol, independent of what computes: route distinct ids to their owners, bring rows back, send
    gradients the reverse way.  `gather_rows(local_rows) -> rows` and `apply_grads(local_rows, grads)` run on the owner."""

    def __init__(self, comm: Comm):
        self.comm = comm
        self.send_counts: List[int] = []
        self.recv_counts: List[int] = []
        self.recv_rows: Optional[torch.Tensor] = None

    def request_rows(self, send_rows: torch.Tensor, send_counts: Sequence[int],
                     gather_rows: Callable[[torch.Tensor], Sequence[torch.Tensor]]) -> List[torch.Tensor]:
        c = self.comm
        self.send_counts = list(send_counts)
        self.recv_counts = c.exchange_counts(send_counts)
        self.recv_rows = c.all_to_all(send_rows, self.send_counts, self.recv_counts)
        answers = gather_rows(self.recv_rows)
        return [c.all_to_all(a, self.recv_counts, self.send_counts) for a in answers]

    def return_grads(self, grads: Sequence[torch.Tensor], apply_grads: Callable[..., None]) -> None:
        c = self.comm
        back = [c.all_to_all(g, self.send_counts, self.recv_counts) for g in grads]
        apply_grads(self.recv_rows, *back)


class ShardedTrainer:
    """One rank of the row-sharded trainer.  `workload` carries the reference flags (model, field_size, feature_size,
    embedding_size, batch (per rank), deep_layers, dropout, l2_reg, learning_rate, optimizer, cross_layers)."""

    def __init__(self, workload: Dict, rank: int, world: int, device: torch.device, table_mode: str = "dense_exact",
                 seed: int = 1, group=None, init_scale: float = 0.01, params: Optional[Dict[str, np.ndarray]] = None):
        self.w = dict(workload)
        self.rank, self.world, self.dev = rank, world, device
        self.comm = Comm(group)
        assert self.comm.world == world and self.comm.rank == rank
        self.x = ShardExchange(self.comm)
        w = self.w
        self.F, self.K, self.V, self.B = w["field_size"], w["embedding_size"], w["feature_size"], w["batch"]
        self.has_lin = w["model"] != "dcn"
        self.eng = Engine(EngineConfig(model=w["model"], field_size=self.F, feature_size=self.V, embedding_size=self.K,
                                       deep_layers=w["deep_layers"], dropout=w["dropout"], cross_layers=w.get("cross_layers", 3),
                                       l2_reg=w["l2_reg"], learning_rate=w["learning_rate"], optimizer=w["optimizer"],
                                       table_mode=table_mode, max_batch=self.B, seed=seed, shard_rank=rank, shard_world=world,
                                       use_graph=False))
        self._lib = capi.lib()
        self._h = self.eng._h
        # requester-side grouping over the GLOBAL id space
        self._g = C.c_void_p()
        cap = self.B * self.F
        capi.check(self._lib.dctr_group_create(self.V, cap, self.K, C.byref(self._g)))
        bufs = [C.c_void_p() for _ in range(8)]
        capi.check(self._lib.dctr_group_buffers(self._g, *[C.byref(b) for b in bufs]))
        self._g_gemb, self._g_glin = bufs[6], bufs[7]
        i32 = dict(dtype=torch.int32, device=device)
        f32 = dict(dtype=torch.float32, device=device)
        self.send_rows = torch.empty(cap, **i32)
        self.upos = torch.empty(cap, **i32)
        self.counts = torch.zeros(2 * world, **i32)
        self.idx = torch.empty(cap, **i32)
        self.send_gemb = torch.empty(cap, self.K, **f32)
        self.send_glin = torch.empty(cap, **f32)
        self.init_params(params, init_scale, seed)

    # -- parameters: every rank draws the same full tensors and keeps its rows -----------------------------------
    def init_params(self, params: Optional[Dict[str, np.ndarray]], scale: float, seed: int) -> None:
        rng = np.random.default_rng(seed)
        full_rows = self.V
        for name, shp in self.eng.param_shapes.items():
            if name in ("emb", "linear"):
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

    # -- owner-side callbacks ----------------------------------------------------------------------------------------
    def _gather_rows(self, rows: torch.Tensor):
        n = int(rows.shape[0])
        st = capi.current_stream()
        emb = torch.empty(max(n, 1), self.K, dtype=torch.float32, device=self.dev)[:n]
        lin = torch.empty(max(n, 1), dtype=torch.float32, device=self.dev)[:n] if self.has_lin else None
        capi.check(self._lib.dctr_table_gather_rows(self._h, capi.ptr(rows), n, capi.ptr(emb), capi.ptr(lin), st))
        return [emb, lin] if self.has_lin else [emb]

    def _apply_grads(self, rows: torch.Tensor, gemb: torch.Tensor, glin: Optional[torch.Tensor] = None):
        n = int(rows.shape[0])
        capi.check(self._lib.dctr_table_apply_grads(self._h, capi.ptr(rows), n, capi.ptr(gemb), capi.ptr(glin),
                                                    capi.current_stream()))

    # -- one synchronous step -------------------------------------------------------------------------------------------
    def _forward_backward(self, ids, vals, labels, train: bool):
        L, st = self._lib, capi.current_stream()
        B = int(ids.shape[0])
        n = B * self.F
        W = self.world
        capi.check(L.dctr_group_ids(self._g, capi.ptr(ids), B, self.F, st))
        capi.check(L.dctr_route_unique(self._g, W, capi.ptr(self.send_rows), capi.ptr(self.upos), capi.ptr(self.counts), st))
        send_counts = [int(x) for x in self.counts[:W].cpu()]          # the step's one host sync: all-to-all split sizes
        U = sum(send_counts)
        back = self.x.request_rows(self.send_rows[:U], send_counts, self._gather_rows)
        rows_back = back[0]
        lin_back = back[1] if self.has_lin else None
        capi.check(L.dctr_entry_index(self._g, capi.ptr(ids), n, capi.ptr(self.upos), capi.ptr(self.idx), st))
        capi.check(L.dctr_step_begin(self._h, st)) if train else None
        capi.check(L.dctr_sharded_forward_backward(self._h, capi.ptr(rows_back), capi.ptr(lin_back), U, capi.ptr(self.idx),
                                                   capi.ptr(vals), capi.ptr(labels), B, B * W, int(train), st))
        return U

    def train_step(self, ids, vals, labels, want_loss: bool = False) -> Optional[float]:
        L, st = self._lib, capi.current_stream()
        B = int(ids.shape[0])
        U = self._forward_backward(ids, vals, labels, True)
        # sparse side: per-distinct-id gradients -> send order -> owners
        capi.check(L.dctr_sharded_row_grads(self._h, self._g, B, st))
        capi.check(L.dctr_permute_unique_rows(self._g, self._g_gemb, capi.ptr(self.upos), self.K, capi.ptr(self.send_gemb), st))
        grads = [self.send_gemb[:U]]
        if self.has_lin:
            capi.check(L.dctr_permute_unique_rows(self._g, self._g_glin, capi.ptr(self.upos), 1, capi.ptr(self.send_glin), st))
            grads.append(self.send_glin[:U])
        self.x.return_grads(grads, self._apply_grads)
        # dense side: flat gradient arena -> all-reduce (sum; the logit gradient already carries 1/global_batch) -> optimizer
        flat, nflat = C.c_void_p(), C.c_int64()
        capi.check(L.dctr_dense_grads(self._h, C.byref(flat), C.byref(nflat), st))
        g = _as_tensor(flat.value, nflat.value, self.dev)
        self.comm.all_reduce_sum(g)
        capi.check(L.dctr_dense_apply(self._h, st))
        if not want_loss:
            return None
        sc = (C.c_float * 4)()
        capi.check(L.dctr_read_scalars(self._h, C.byref(sc), st))
        t = torch.tensor([sc[0], sc[1], sc[2]], dtype=torch.float64, device=self.dev)
        self.comm.all_reduce_sum(t)
        xent, sq_emb, sq_lin = (float(v) for v in t.cpu())
        return xent / (B * self.world) + self.w["l2_reg"] * 0.5 * (sq_emb + sq_lin + float(sc[3]))

    def predict(self, ids, vals) -> torch.Tensor:
        self._forward_backward(ids, vals, None, False)
        p = C.c_void_p()
        capi.check(self._lib.dctr_last_outputs(self._h, C.byref(p), None))
        return _as_tensor(p.value, int(ids.shape[0]), self.dev).clone()

    def close(self):
        if self._g is not None and self._g.value:
            self._lib.dctr_group_destroy(self._g)
            self._g = C.c_void_p()
        self.eng.close()


def _as_tensor(ptr: int, n: int, dev: torch.device) -> torch.Tensor:
    """Zero-copy torch view of `n` floats of engine-owned device memory (for RCCL calls)."""
    class _Arr:
        pass
    a = _Arr()
    a.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}
    return torch.as_tensor(a, device=dev)
