"""Row records (csrc/engine.h tab_ld; include/deepctr_hip.h dctr_param_device_view): a handle whose table rows may lag keeps each row of
`emb` / `linear` next to its two Adam slots in one record.  The layout is storage only -- every kernel that touches a table row takes
the row strides -- so the SAME steps on a handle created with DCTR_TABLE_RECORDS=0 (six dense arrays, the layout every other handle
has) must give the same tables and slots: the arithmetic per row is identical, only addresses differ.

Covered here on the lagging path (sweep period 4, hints, loss-reporting steps in between: the catch-up gather, the touched-rows step,
the background sweep, the flush and its sum theta^2, eval's l2 term), the accessors (get / set / slots through the strided view), the
device view itself, and the models without a linear table (DCN: records of 3 K floats)."""
import os

import numpy as np
import pytest
import torch

from tests.util import dev_batch
from tf_repos_amd.engine import Engine, EngineConfig
from tf_repos_amd.synth import synth_batch

pytestmark = pytest.mark.gpu

F, V, B = 13, 30_000, 512


def _same(a, b):
    """Same arithmetic per row; the long id segments of the fused tail fold through atomics, whose order is not fixed: last-bit room."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max()) <= 1e-6 * max(float(np.abs(b).max()), 1e-30)


def _make(records, **kw):
    old = os.environ.get("DCTR_TABLE_RECORDS")
    os.environ["DCTR_TABLE_RECORDS"] = "1" if records else "0"
    try:
        return Engine(EngineConfig(field_size=F, feature_size=V, max_batch=B, seed=3, l2_reg=1e-3, learning_rate=1e-3, optimizer="Adam",
                                   table_sweep_period=4, use_graph=False, **kw))
    finally:
        if old is None:
            del os.environ["DCTR_TABLE_RECORDS"]
        else:
            os.environ["DCTR_TABLE_RECORDS"] = old


@pytest.mark.parametrize("model,K,extra", [("deepfm", 16, {}), ("deepfm", 8, {}), ("dcn", 16, {"cross_layers": 2}), ("nfm", 32, {})])
def test_records_and_dense_arrays_train_alike(model, K, extra, dev):
    kw = dict(model=model, embedding_size=K, deep_layers=(64, 32), dropout=(0.8, 0.8), **extra)
    rec, flat = _make(True, **kw), _make(False, **kw)
    assert not rec.param_tensor("emb").is_contiguous() and flat.param_tensor("emb").is_contiguous()
    rng = np.random.default_rng(11)
    init = {n: rng.normal(0, 0.05, size=s).astype(np.float32) for n, s in rec.param_shapes.items()}
    for e in (rec, flat):
        e.set_params({n: torch.from_numpy(a) for n, a in init.items()})
    for n, a in init.items():           # set -> get through the strided view
        assert np.array_equal(rec.get_param(n), a), n
    batches = [dev_batch(*synth_batch(B, F, V, seed=500 + i), dev) for i in range(14)]
    for i, b in enumerate(batches):
        want = i in (5, 9)              # (a loss-reporting step flushes every row first)
        lr, lf = rec.train_step(*b, want_loss=want), flat.train_step(*b, want_loss=want)
        if want:
            assert abs(lr - lf) <= 1e-6 * max(1.0, abs(lf)), (i, lr, lf)
        if i + 1 < len(batches) and i % 3 != 2:
            rec.prefetch_ids(batches[i + 1][0]); flat.prefetch_ids(batches[i + 1][0])
    # the device view, row by row, against the host copy (rows lag until something asks for the table: the view does)
    rows = torch.from_numpy(np.unique(synth_batch(B, F, V, seed=500)[0])[:2000]).to(dev).long()
    view = rec.param_tensor("emb")
    assert view.shape == (V, K) and view.stride(0) > K and view.stride(1) == 1
    got_r, got_f = rec.get_params(), flat.get_params()
    assert np.array_equal(view[rows].cpu().numpy(), got_r["emb"][rows.cpu().numpy()])
    for n in got_f:
        assert _same(got_r[n], got_f[n]), (n, float(np.abs(got_r[n] - got_f[n]).max()))
    for n in ("emb", "linear"):
        if n in rec.param_shapes:
            for which in (0, 1):
                assert _same(rec.get_slot(n, which), flat.get_slot(n, which)), (n, which)
    # predict and eval (their gather reads the records; eval's loss sums theta^2 over the strided variables)
    ev = batches[0]
    out = []
    for e in (rec, flat):
        prob = torch.empty(B, device=dev)
        e.predict(ev[0], ev[1], out_prob=prob)
        e.eval_reset()
        e.eval_batch(*ev)
        out.append((prob.cpu().numpy(), e.eval_result()))
    assert _same(out[0][0], out[1][0])
    assert abs(out[0][1][1] - out[1][1][1]) <= 1e-6 * max(1.0, abs(out[1][1][1])), (out[0][1], out[1][1])
    rec.close(); flat.close()


def test_slot_write_through_the_view_round_trips(dev):
    rec = _make(True, model="deepfm", embedding_size=16, deep_layers=(32,), dropout=(1.0,))
    rng = np.random.default_rng(5)
    for n in ("emb", "linear"):
        for which in (0, 1):
            a = np.abs(rng.normal(0, 1e-3, size=rec.param_shapes[n])).astype(np.float32)
            rec.set_slot(n, which, a)
            assert np.array_equal(rec.get_slot(n, which), a), (n, which)
    # theta untouched by the slot writes (the record's three groups do not overlap)
    assert float(np.abs(rec.get_param("linear")).max()) == 0.0 and float(np.abs(rec.get_param("emb")).max()) == 0.0
    import ctypes as C
    from tf_repos_amd import capi
    p = C.c_void_p()
    assert rec._lib.dctr_param_device_ptr(rec._h, b"emb", C.byref(p)) != 0      # the plain pointer entry refuses a strided variable
    assert rec._lib.dctr_param_device_ptr(rec._h, b"mlp0/weights", C.byref(p)) == 0
    rec.close()
