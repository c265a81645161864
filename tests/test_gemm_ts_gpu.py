"""Tall split-precision products (csrc/gemm_ts.h, round 6): AFM's attention layer over the B * P pair rows and its input gradient
(AFM.py:142-147: fully_connected(relu) over [B * P, K], the (A -> 1) score, and their gradients: forward + score, gated input gradient,
gated weight gradient with the two bias / score-weight sums) on the bf16 matrix pipe with every f32
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
    sbuf = torch.full((M * 4 + guard,), -1, dtype=torch.int64, device=dev)          # the sign words of the output, 32 bytes per row
    capi.check(lib.dctr_fc_fwd_dot_split(capi.ptr(dx_), K, capi.ptr(dw_), capi.ptr(db_), capi.ptr(ybuf), N, M, K, N, capi.ptr(dwo_), capi.ptr(dbuf),
                                         capi.ptr(sbuf), capi.ptr(ws), st))
    capi.check(lib.dctr_fc_fwd(capi.ptr(dx_), K, capi.ptr(dw_), capi.ptr(db_), capi.ptr(ye), N, M, K, N, 1, 1.0, 0, st))
    ys = ybuf[:M * N].view(M, N).cpu()
    assert bool(torch.isnan(ybuf[M * N:]).all()) and bool(torch.isnan(dbuf[M:]).all()) and bool((sbuf[M * 4:] == -1).all()), "stores past the last row"
    es, ee = float((ys.double() - ref).abs().max()), float((ye.cpu().double() - ref).abs().max())
    print("fwd   %6d x %3d x %3d: split max err %.2e, exact %.2e" % (M, K, N, es, ee))
    # the sign words: bit 4 tt + r of word (row, q) = 1[y[row, 16 tt + 4 q + r] > 0]
    pos = (ys > 0).view(M, N // 16, 4, 4).permute(0, 2, 1, 3).reshape(M, 4, N // 4).to(torch.int64)     # [row, q, 4 tt + r]
    want = (pos << torch.arange(N // 4, dtype=torch.int64)).sum(-1)
    assert torch.equal(sbuf[:M * 4].view(M, 4).cpu(), want)
    assert es <= 2 * ee + 1e-9, (es, ee)
    dref = ys.double() @ wo.double()
    ed = float((dbuf[:M].cpu().double() - dref).abs().max())
    e32 = float(((ys.to(dev) @ dwo_).cpu().double() - dref).abs().max())
    print("score dot: max err %.2e (an f32 matvec of the same rows: %.2e)" % (ed, e32))
    assert ed <= 2 * e32 + 1e-7, (ed, e32)
    # twice the same bits (LDS-DMA staging, barriers and two waves per SIMD: a schedule-dependent result would be a race)
    yb2 = torch.empty(M, N, device=dev); db2 = torch.empty(M, device=dev)
    for _ in range(3):
        capi.check(lib.dctr_fc_fwd_dot_split(capi.ptr(dx_), K, capi.ptr(dw_), capi.ptr(db_), capi.ptr(yb2), N, M, K, N, capi.ptr(dwo_), capi.ptr(db2), None, capi.ptr(ws), st))
        assert torch.equal(yb2.cpu(), ys) and torch.equal(db2, dbuf[:M])
    # without the dot output
    y2 = torch.empty(M, N, device=dev)
    capi.check(lib.dctr_fc_fwd_dot_split(capi.ptr(dx_), K, capi.ptr(dw_), capi.ptr(db_), capi.ptr(y2), N, M, K, N, capi.ptr(dwo_), None, None, capi.ptr(ws), st))
    assert torch.equal(y2.cpu(), ys)
    # ---- gated input gradient: dX = (rs (x) wo . 1[H > 0]) W^T, H = the forward's output
    h = ys.to(dev)
    wk = (w * wo).double()                                   # (the kernels round W[k, a] wo[a] to f32 once, like this product)
    ref = ((ys > 0).double() @ wk.t()) * rs.double()[:, None]
    gbuf = torch.full((M * K + guard,), float("nan"), device=dev)
    capi.check(lib.dctr_fc_bwd_data_gate_split(capi.ptr(h), N, None, capi.ptr(drs_), capi.ptr(dwo_), capi.ptr(dw_), capi.ptr(gbuf), K, M, K, N, capi.ptr(ws), st))
    assert bool(torch.isnan(gbuf[M * K:]).all()), "stores past the last row"
    gs = gbuf[:M * K].view(M, K).cpu()
    g2 = torch.empty(M, K, device=dev)
    for _ in range(3):
        capi.check(lib.dctr_fc_bwd_data_gate_split(capi.ptr(h), N, None, capi.ptr(drs_), capi.ptr(dwo_), capi.ptr(dw_), capi.ptr(g2), K, M, K, N, capi.ptr(ws), st))
        assert torch.equal(g2.cpu(), gs)
    # ... and from the forward's sign words instead of the rows (two column halves, two blocks per CU): the same bits, run after run
    gb = torch.full((M * K + guard,), float("nan"), device=dev)
    for _ in range(3):
        capi.check(lib.dctr_fc_bwd_data_gate_split(None, 0, capi.ptr(sbuf), capi.ptr(drs_), capi.ptr(dwo_), capi.ptr(dw_), capi.ptr(gb), K, M, K, N, capi.ptr(ws), st))
        assert torch.equal(gb[:M * K].view(M, K).cpu(), gs) and bool(torch.isnan(gb[M * K:]).all())
    g32 = (((h > 0).float() @ (dw_ * dwo_).t()) * drs_[:, None]).cpu()
    es, ee = float((gs.double() - ref).abs().max()), float((g32.double() - ref).abs().max())
    print("gate  %6d x %3d x %3d: split max err %.2e, an f32 product %.2e" % (M, K, N, es, ee))
    assert es <= 2 * ee + 1e-9, (es, ee)
    # ---- gated weight gradient: dW = X^T (rs (x) wo . 1[H > 0]), db = its column sums, dwo = sum_r rs[r] H[r, :]
    gate = (ys > 0).double()
    xs = x.double() * rs.double()[:, None]
    ref_w = (xs.t() @ gate) * wo.double()[None, :]
    ref_b = (gate * rs.double()[:, None]).sum(0) * wo.double()
    ref_o = (ys.double() * rs.double()[:, None]).sum(0)
    per = K * N + 2 * N
    wsp = torch.zeros(256 * per, device=dev)
    dw, db, dwo = torch.empty(K, N, device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev)
    capi.check(lib.dctr_fc_bwd_weights_gate_split(capi.ptr(dx_), K, capi.ptr(h), N, capi.ptr(drs_), capi.ptr(dwo_), capi.ptr(dw), capi.ptr(db), capi.ptr(dwo),
                                                  M, K, N, capi.ptr(wsp), wsp.numel() * 4, st))
    d32 = (h > 0).float() * drs_[:, None] * dwo_[None, :]                       # the gradient the exact path forms on its operand loads
    w32 = torch.empty(K, N, device=dev); b32 = torch.empty(N, device=dev)
    wse = torch.zeros(64 * (K * N + N), device=dev)
    capi.check(lib.dctr_fc_bwd_weights(capi.ptr(dx_), K, capi.ptr(d32), N, capi.ptr(w32), capi.ptr(b32), M, K, N, capi.ptr(wse), wse.numel() * 4, st))
    es, ee = float((dw.cpu().double() - ref_w).abs().max()), float((w32.cpu().double() - ref_w).abs().max())
    print("wgrad %6d x %3d x %3d: split max err %.2e, exact (on the materialised gradient) %.2e, values to %.1f" % (M, K, N, es, ee, float(ref_w.abs().max())))
    assert es <= 2 * ee + 1e-9, (es, ee)
    eb, eeb = float((db.cpu().double() - ref_b).abs().max()), float((b32.cpu().double() - ref_b).abs().max())
    assert eb <= 2 * eeb + 1e-6 * float(ref_b.abs().max()), (eb, eeb)
    eo = float((dwo.cpu().double() - ref_o).abs().max())
    o32 = float(((h * drs_[:, None]).sum(0).cpu().double() - ref_o).abs().max())
    assert eo <= 2 * o32 + 1e-6 * float(ref_o.abs().max()), (eo, o32)
    # a workspace of ONE slab: the same sums from a single block
    capi.check(lib.dctr_fc_bwd_weights_gate_split(capi.ptr(dx_[:4096]), K, capi.ptr(h[:4096]), N, capi.ptr(drs_), capi.ptr(dwo_), capi.ptr(dw), capi.ptr(db), capi.ptr(dwo),
                                                  4096, K, N, capi.ptr(wsp), per * 4, st))
    ref1 = (xs[:4096].t() @ gate[:4096]) * wo.double()[None, :]
    assert float((dw.cpu().double() - ref1).abs().max()) <= 1e-5 * max(1.0, float(ref1.abs().max()))


@pytest.mark.parametrize("B,F,K,N", [(96, 39, 256, 256), (300, 23, 128, 256), (130, 39, 256, 128)])
def test_rows_formed_from_the_embeddings_give_the_same_bits(B, F, K, N, dev):
    """AFM.py:130-139's element-wise products e_i . e_j as the tall operand WITHOUT the [B P, K] tensor: the forward product and the gated weight
    gradient read the gathered embeddings and multiply in the registers -- every output word equal to the op over the materialised rows."""
    lib = capi.lib()
    st = capi.current_stream()
    P = F * (F - 1) // 2
    M = B * P - 5                                            # (the last example's last pairs are not rows: a ragged end)
    g = torch.Generator().manual_seed(B + F + K + N)
    e = (torch.rand(B, F * K, generator=g) * 2 - 1).to(dev)
    pi = torch.tensor([i for i in range(F - 1) for j in range(i + 1, F)], dtype=torch.int16)
    pj = torch.tensor([j for i in range(F - 1) for j in range(i + 1, F)], dtype=torch.int16)
    dpi, dpj = pi.to(dev), pj.to(dev)
    ev = e.view(B, F, K)
    x = (ev[:, pi.long(), :] * ev[:, pj.long(), :]).reshape(B * P, K)[:M].contiguous()
    w = ((torch.rand(K, N, generator=g) * 2 - 1) * 0.08).to(dev)
    b = ((torch.rand(N, generator=g) * 2 - 1) * 0.1).to(dev)
    wo = ((torch.rand(N, generator=g) * 2 - 1) * 0.3).to(dev)
    rs = (torch.rand(M, generator=g) * 2 - 1).to(dev)
    ws = _ws(lib, 256, 256, dev)
    y1, y2 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    d1, d2 = torch.empty(M, device=dev), torch.empty(M, device=dev)
    s1, s2 = torch.zeros(M * 4, dtype=torch.int64, device=dev), torch.zeros(M * 4, dtype=torch.int64, device=dev)
    capi.check(lib.dctr_fc_fwd_dot_split(capi.ptr(x), K, capi.ptr(w), capi.ptr(b), capi.ptr(y1), N, M, K, N, capi.ptr(wo), capi.ptr(d1), capi.ptr(s1), capi.ptr(ws), st))
    capi.check(lib.dctr_pairs_fc_fwd_dot_split(capi.ptr(e), F * K, B, capi.ptr(dpi), capi.ptr(dpj), P, capi.ptr(w), capi.ptr(b), capi.ptr(y2), N, M, K, N,
                                               capi.ptr(wo), capi.ptr(d2), capi.ptr(s2), capi.ptr(ws), st))
    assert torch.equal(y1, y2) and torch.equal(d1, d2) and torch.equal(s1, s2)
    per = K * N + 2 * N
    wsp = torch.zeros(256 * per, device=dev)
    out1 = [torch.empty(K, N, device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev)]
    out2 = [torch.empty(K, N, device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev)]
    capi.check(lib.dctr_fc_bwd_weights_gate_split(capi.ptr(x), K, capi.ptr(y1), N, capi.ptr(rs), capi.ptr(wo), *[capi.ptr(t) for t in out1], M, K, N,
                                                  capi.ptr(wsp), wsp.numel() * 4, st))
    capi.check(lib.dctr_pairs_fc_bwd_weights_gate_split(capi.ptr(e), F * K, B, capi.ptr(dpi), capi.ptr(dpj), P, capi.ptr(y1), N, None, None, None, capi.ptr(rs),
                                                        capi.ptr(wo), *[capi.ptr(t) for t in out2], M, K, N, capi.ptr(wsp), wsp.numel() * 4, st))
    for a, b2 in zip(out1, out2):
        assert torch.equal(a, b2)
    # ... and without reading the layer's output at all: the gate from the forward's sign words (same dW and db, bit for bit), the second column
    # sums as sum_k W[k, n] dWraw[k, n] + b[n] dbraw[n] (the same number in another summation order) -- and a forward that stores only scores + signs
    out3 = [torch.empty(K, N, device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev)]
    capi.check(lib.dctr_pairs_fc_bwd_weights_gate_split(capi.ptr(e), F * K, B, capi.ptr(dpi), capi.ptr(dpj), P, None, 0, capi.ptr(s1), capi.ptr(w), capi.ptr(b),
                                                        capi.ptr(rs), capi.ptr(wo), *[capi.ptr(t) for t in out3], M, K, N, capi.ptr(wsp), wsp.numel() * 4, st))
    assert torch.equal(out1[0], out3[0]) and torch.equal(out1[1], out3[1])
    out4 = [torch.empty(K, N, device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev)]
    for _ in range(3):                                   # (the conversion of the next rows runs between the MFMAs of these: same bits every run)
        capi.check(lib.dctr_pairs_fc_bwd_weights_gate_split(capi.ptr(e), F * K, B, capi.ptr(dpi), capi.ptr(dpj), P, None, 0, capi.ptr(s1), capi.ptr(w), capi.ptr(b),
                                                            capi.ptr(rs), capi.ptr(wo), *[capi.ptr(t) for t in out4], M, K, N, capi.ptr(wsp), wsp.numel() * 4, st))
        assert all(torch.equal(a, c) for a, c in zip(out3, out4))
    ref_o = (y1.double() * rs.double()[:, None]).sum(0)
    e1, e3 = float((out1[2].double() - ref_o).abs().max()), float((out3[2].double() - ref_o).abs().max())
    print("second column sums: over H max err %.2e, from the product %.2e (values to %.1f)" % (e1, e3, float(ref_o.abs().max())))
    assert e3 <= 4 * e1 + 2e-6 * float(ref_o.abs().max()), (e1, e3)
    d3, s3 = torch.empty(M, device=dev), torch.zeros(M * 4, dtype=torch.int64, device=dev)
    capi.check(lib.dctr_pairs_fc_fwd_dot_split(capi.ptr(e), F * K, B, capi.ptr(dpi), capi.ptr(dpj), P, capi.ptr(w), capi.ptr(b), None, N, M, K, N,
                                               capi.ptr(wo), capi.ptr(d3), capi.ptr(s3), capi.ptr(ws), st))
    assert torch.equal(d3, d1) and torch.equal(s3, s1)
    # more rows than the embeddings hold pairs for: refused
    rc = lib.dctr_pairs_fc_fwd_dot_split(capi.ptr(e), F * K, B - 1, capi.ptr(dpi), capi.ptr(dpj), P, capi.ptr(w), capi.ptr(b), capi.ptr(y2), N, M, K, N,
                                         capi.ptr(wo), capi.ptr(d2), None, capi.ptr(ws), st)
    assert rc == capi.DCTR_ERR_UNSUPPORTED


def test_shapes_not_taken_are_refused(dev):
    lib = capi.lib()
    st = capi.current_stream()
    ws = _ws(lib, 256, 256, dev)
    for M, K, N in [(4096, 256, 256), (70000, 64, 256), (70000, 256, 192)]:
        x = torch.zeros(M, K, device=dev); w = torch.zeros(K, N, device=dev); b = torch.zeros(N, device=dev); y = torch.zeros(M, N, device=dev)
        rc = lib.dctr_fc_fwd_dot_split(capi.ptr(x), K, capi.ptr(w), capi.ptr(b), capi.ptr(y), N, M, K, N, capi.ptr(b), None, None, capi.ptr(ws), st)
        assert rc == capi.DCTR_ERR_UNSUPPORTED, (M, K, N, rc)
        r = torch.zeros(M, device=dev)
        rc = lib.dctr_fc_bwd_data_gate_split(capi.ptr(y), N, None, capi.ptr(r), capi.ptr(b), capi.ptr(w), capi.ptr(x), K, M, K, N, capi.ptr(ws), st)
        assert rc == capi.DCTR_ERR_UNSUPPORTED, (M, K, N, rc)
