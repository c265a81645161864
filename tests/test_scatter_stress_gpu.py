"""Stress of the step's fused tail (csrc/group.hip scatter_apply_kernel, C ABI dctr_embed_scatter_apply) on the inputs that take its
cross-block path: segments of >= 256 entries -- Criteo's 13 numeric ids occur in EVERY example (get_criteo_feature.py:138-145), so
that path runs every step of every reference workload -- are cut into 256-entry chunks dealt to different blocks; the partial sums
meet in a compact row through returned float atomics, each chunk adds its entry count to a completion ticket, and the block that
completes the segment takes the total out with atomicExch (leaving zeros for the next batch) and steps the table row.  No fence
orders the adds before the ticket: the kernel relies on a returned atomic having been PERFORMED at the device-scope point of
coherence before the ticket is drawn.  That is how gfx950 behaves, not something the memory model spells out -- so this test runs
the path 20 000 times and checks EVERY row of EVERY step against an fp64 segment sum (UnsortedSegmentSum, the gradient of
DeepFM.py:126,130's gathers): a ticket that overtakes an add loses (or delays into the next step) a 256-entry partial sum, ~16 in
magnitude here against a tolerance of ~1e-2.

The optimizer is Momentum with momentum 0 and lr 0 (opt_rules.h: accum = 0 * accum + g; theta -= 0 * accum): the row's first
slot then holds exactly the gradient the step saw, theta never moves, and the check is a device-side max over the whole table."""
import ctypes as C

import numpy as np
import pytest
import torch

from tf_repos_amd import capi
from tf_repos_amd.synth import synth_batch

pytestmark = pytest.mark.gpu

MOMENTUM = capi.OPTIMIZERS["Momentum"]


def _dense_segment_sum(ids, grad_rows, V):
    """fp64 [V, C] sum of grad_rows [n, C] over equal ids [n], and the same over |grad_rows| (the yardstick of the tolerance)"""
    out = np.zeros((V, grad_rows.shape[1]), np.float64)
    mag = np.zeros((V, grad_rows.shape[1]), np.float64)
    np.add.at(out, ids, grad_rows.astype(np.float64))
    np.add.at(mag, ids, np.abs(grad_rows.astype(np.float64)))
    return out, mag


def _run(dev, ids, vals, V, K, steps, n_variants=4, seed=0):
    B, F = ids.shape
    lib = capi.lib()
    st = capi.current_stream()
    g = C.c_void_p()
    capi.check(lib.dctr_group_create(V, B * F, K, C.byref(g)))
    rng = np.random.default_rng(seed)
    d_ids = torch.from_numpy(ids).to(dev)
    d_vals = torch.from_numpy(vals).to(dev)
    variants = []
    flat_ids = ids.reshape(-1)
    for _ in range(n_variants):
        dE = rng.normal(0, 1, size=(B, F * K)).astype(np.float32)
        dy = rng.normal(0, 1, size=B).astype(np.float32)
        rows = (dE.reshape(B, F, K) * vals[:, :, None]).reshape(B * F, K)
        exp_e, mag_e = _dense_segment_sum(flat_ids, rows, V)
        exp_l, mag_l = _dense_segment_sum(flat_ids, (dy[:, None] * vals).reshape(B * F, 1), V)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.float32))).to(dev)
        variants.append(dict(dE=t(dE), dy=t(dy), exp_e=t(exp_e), exp_l=t(exp_l[:, 0]),
                             tol_e=t(2e-6 * mag_e + 1e-6), tol_l=t(2e-6 * mag_l[:, 0] + 1e-6)))
    emb = torch.full((V, K), 0.25, device=dev)
    lin = torch.full((V,), -0.5, device=dev)
    s0 = torch.zeros(V, K, device=dev)
    l0 = torch.zeros(V, device=dev)
    hyper = np.array([0.0, 0.0, 0, 0, 0, 0, 0, 0], np.float32)          # lr 0, momentum 0: slot0 <- the gradient, theta unchanged
    worst = torch.zeros((), device=dev)                                 # max over steps of (|error| / tolerance)
    for step in range(steps):
        v = variants[step % n_variants]
        capi.check(lib.dctr_group_ids(g, capi.ptr(d_ids), B, F, st))
        capi.check(lib.dctr_embed_scatter_apply(g, MOMENTUM, capi.ptr(hyper), capi.ptr(emb), capi.ptr(s0), None, capi.ptr(lin), capi.ptr(l0),
                                                None, 0.0, None, capi.ptr(v["dE"]), F * K, None, 0, None, None, capi.ptr(v["dy"]),
                                                capi.ptr(d_vals), B, F, K, capi.GATHER_RAW, st))
        r = torch.maximum(((s0 - v["exp_e"]).abs() / v["tol_e"]).max(), ((l0 - v["exp_l"]).abs() / v["tol_l"]).max())
        worst = torch.maximum(worst, r)
    worst = float(worst)
    # the kernel leaves what the next grouping relies on: compact rows and slot words back at zero, theta untouched
    bufs = [C.c_void_p() for _ in range(8)]
    capi.check(lib.dctr_group_buffers(g, *[C.byref(b) for b in bufs]))
    n = B * F
    gemb = np.empty((n, K), np.float32)
    slot = np.empty(V, np.int32)
    capi.check(lib.dctr_memcpy_d2h(capi.ptr(gemb), bufs[6], gemb.nbytes, st))
    capi.check(lib.dctr_memcpy_d2h(capi.ptr(slot), bufs[4], slot.nbytes, st))
    torch.cuda.synchronize()
    capi.check(lib.dctr_group_destroy(g))
    assert not gemb.any() and not slot.any()
    assert float((emb - 0.25).abs().max()) == 0.0 and float((lin + 0.5).abs().max()) == 0.0
    return worst


def test_one_id_in_every_entry(dev):
    """all_same_id: B x F = 159 744 entries of ONE row -- 624 chunks over 256 blocks, one ticket"""
    B, F, V, K = 4096, 39, 1000, 16
    ids = np.full((B, F), 7, np.int32)
    vals = np.random.default_rng(1).random((B, F)).astype(np.float32)
    worst = _run(dev, ids, vals, V, K, steps=6000)
    assert worst <= 1.0, "error / tolerance = %.3g" % worst


def test_criteo_shape_thirteen_hot_ids(dev):
    """the reference's own shape: ids 1..13 in every example (13 long segments of B entries, 16 chunks each, interleaved over the
    blocks), Zipf-headed categorical fields behind them (medium segments by the wave path, the tail by the walkers)"""
    B, F, V, K = 4096, 39, 200_000, 16
    ids, vals, _ = synth_batch(B, F, V, seed=5)
    worst = _run(dev, ids, vals, V, K, steps=14000)
    assert worst <= 1.0, "error / tolerance = %.3g" % worst


@pytest.mark.parametrize("K", [8, 64])
def test_other_row_widths(K, dev):
    """long_segment(K/4) differs per width (group.hip): 256 entries up to K = 32, 128 at K = 64"""
    B, F, V = 1024, 39, 50_000
    ids, vals, _ = synth_batch(B, F, V, seed=6)
    worst = _run(dev, ids, vals, V, K, steps=300)
    assert worst <= 1.0, "error / tolerance = %.3g" % worst
