"""Tall split-precision products (csrc/gemm_ts.h, round 6): AFM's attention layer over the B * P pair rows and its input gradient
(AFM.py:142-147: fully_connected(relu) over [B * P, K], the (A -> 1) score, and their gradients) on the bf16 matrix pipe with every f32
value as three bf16 planes -- what an AFM handle at the reference's K = 256 (run.sh:18) runs by default (gemm_mode split).

Checked through the C ABI against an fp64 product of the same f32 inputs, on ragged row counts (the last block tile, wave and 16-row
fragment are all partial): the split kernel's max error within 2x the exact f32 path's on the same inputs ("f32-equivalent" as a number),
nothing written past the last row, and shapes the kernels do not take refused with DCTR_ERR_UNSUPPORTED.  The engine-level parity of the
AFM step that runs on these kernels is tests/test_fullsize_gpu.py (K = A = 256) and tests/test_afm_grad_gpu.py (K = 256, A = 128)."""
import ctypes as C

import pytest
import torch

from tf_repos_amd import capi

pytestmark = pytest.mark.gpu

SHAPES = [(65573, 256, 256), (100037, 256, 128), (70001, 128, 256), (65536, 128, 128)]


def _ws(lib, R, N, dev):
    nb = C.c_int64()
    capi.check(lib.dctr_ts_plane_bytes(R, N, C.byref(nb)))
    assert nb.value == 3 * R * N * 2
    return torch.zeros(nb.value // 4, dtype=torch.int32, device=dev)


@pytest.mark.parametrize("M,K,N", SHAPES)
def test_tall_products_are_f32_equivalent(M, K, N, dev):
    lib = capi.lib()
    st = capi.current_stream()
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.rand(M, K, generator=g) * 2 - 1
    w = (torch.rand(K, N, generator=g) * 2 - 1) * 0.08
    b = (torch.rand(N, generator=g) * 2 - 1) * 0.1
    wo = (torch.rand(N, generator=g) * 2 - 1) * 0.3
    rs = torch.rand(M, generator=g) * 2 - 1
    dx_, dw_, db_, dwo_, drs_ = x.to(dev), w.to(dev), b.to(dev), wo.to(dev), rs.to(dev)
    ws = _ws(lib, 256, 256, dev)
    guard = 4096
    # ---- forward: relu(x W + b) and the score dot from the same accumulators
    ref = torch.relu(x.double() @ w.double() + b.double())
    ybuf = torch.full((M * N + guard,), float("nan"), device=dev)
    dbuf = torch.full((M + guard,), float("nan"), device=dev)
    ye = torch.empty(M, N, device=dev)
    capi.check(lib.dctr_fc_fwd_dot_split(capi.ptr(dx_), K, capi.ptr(dw_), capi.ptr(db_), capi.ptr(ybuf), N, M, K, N, capi.ptr(dwo_), capi.ptr(dbuf),
                                         capi.ptr(ws), st))
    capi.check(lib.dctr_fc_fwd(capi.ptr(dx_), K, capi.ptr(dw_), capi.ptr(db_), capi.ptr(ye), N, M, K, N, 1, 1.0, 0, st))
    ys = ybuf[:M * N].view(M, N).cpu()
    assert bool(torch.isnan(ybuf[M * N:]).all()) and bool(torch.isnan(dbuf[M:]).all()), "stores past the last row"
    es, ee = float((ys.double() - ref).abs().max()), float((ye.cpu().double() - ref).abs().max())
    print("fwd   %6d x %3d x %3d: split max err %.2e, exact %.2e" % (M, K, N, es, ee))
    assert es <= 2 * ee + 1e-9, (es, ee)
    dref = ys.double() @ wo.double()
    ed = float((dbuf[:M].cpu().double() - dref).abs().max())
    e32 = float(((ys.to(dev) @ dwo_).cpu().double() - dref).abs().max())
    print("score dot: max err %.2e (an f32 matvec of the same rows: %.2e)" % (ed, e32))
    assert ed <= 2 * e32 + 1e-7, (ed, e32)
    # without the dot output
    y2 = torch.empty(M, N, device=dev)
    capi.check(lib.dctr_fc_fwd_dot_split(capi.ptr(dx_), K, capi.ptr(dw_), capi.ptr(db_), capi.ptr(y2), N, M, K, N, capi.ptr(dwo_), None, capi.ptr(ws), st))
    assert torch.equal(y2.cpu(), ys)
    # ---- gated input gradient: dX = (rs (x) wo . 1[H > 0]) W^T, H = the forward's output
    h = ys.to(dev)
    wk = (w * wo).double()                                   # (the kernels round W[k, a] wo[a] to f32 once, like this product)
    ref = ((ys > 0).double() @ wk.t()) * rs.double()[:, None]
    gbuf = torch.full((M * K + guard,), float("nan"), device=dev)
    capi.check(lib.dctr_fc_bwd_data_gate_split(capi.ptr(h), N, capi.ptr(drs_), capi.ptr(dwo_), capi.ptr(dw_), capi.ptr(gbuf), K, M, K, N, capi.ptr(ws), st))
    assert bool(torch.isnan(gbuf[M * K:]).all()), "stores past the last row"
    gs = gbuf[:M * K].view(M, K).cpu()
    g32 = (((h > 0).float() @ (dw_ * dwo_).t()) * drs_[:, None]).cpu()
    es, ee = float((gs.double() - ref).abs().max()), float((g32.double() - ref).abs().max())
    print("gate  %6d x %3d x %3d: split max err %.2e, an f32 product %.2e" % (M, K, N, es, ee))
    assert es <= 2 * ee + 1e-9, (es, ee)


def test_shapes_not_taken_are_refused(dev):
    lib = capi.lib()
    st = capi.current_stream()
    ws = _ws(lib, 256, 256, dev)
    for M, K, N in [(4096, 256, 256), (70000, 64, 256), (70000, 256, 192)]:
        x = torch.zeros(M, K, device=dev); w = torch.zeros(K, N, device=dev); b = torch.zeros(N, device=dev); y = torch.zeros(M, N, device=dev)
        rc = lib.dctr_fc_fwd_dot_split(capi.ptr(x), K, capi.ptr(w), capi.ptr(b), capi.ptr(y), N, M, K, N, capi.ptr(b), None, capi.ptr(ws), st)
        assert rc == capi.DCTR_ERR_UNSUPPORTED, (M, K, N, rc)
        r = torch.zeros(M, device=dev)
        rc = lib.dctr_fc_bwd_data_gate_split(capi.ptr(y), N, capi.ptr(r), capi.ptr(b), capi.ptr(w), capi.ptr(x), K, M, K, N, capi.ptr(ws), st)
        assert rc == capi.DCTR_ERR_UNSUPPORTED, (M, K, N, rc)
