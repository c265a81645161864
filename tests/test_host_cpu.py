"""CPU-side tests (no GPU): the C ABI exports what the header declares, the host parser (K1) matches the oracle and
glibc strtof bit-for-bit, and the oracle's own formulas are self-consistent."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from oracle import deepctr_oracle as O
from tf_repos_amd import capi, errors
from tf_repos_amd.input_pipeline import parse_libsvm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# the only concrete 39-field sample in the reference: deep_fm_serving_client.cpp:42-45
SERVING_IDS = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, 555, 1078, 17797, 26190, 26341, 28570, 35361, 35613, 35984, 48424,
               51364, 64053, 65964, 66206, 71628, 84088, 84119, 86889, 88280, 88283, 100288, 100300, 102447, 109932, 111823]
SERVING_VALS = [0.05, 0.006633, 0.05, 0, 0.021594, 0.008, 0.15, 0.04, 0.362, 0.1, 0.2, 0, 0.04] + [1.0] * 26


def test_library_exports_every_symbol_the_header_declares():
    hdr = open(os.path.join(ROOT, "include", "deepctr_hip.h")).read()
    declared = set(re.findall(r"\b(dctr_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"dctr_config", "dctr_engine", "dctr_group"}
    lib = capi.lib()          # binds every symbol in capi._SIGS or raises
    for name in sorted(declared):
        assert hasattr(lib, name), "header declares %s but libdeepctr_hip.so does not export it" % name
    assert declared == set(capi.DECLARED_SYMBOLS), declared ^ set(capi.DECLARED_SYMBOLS)
    assert lib.dctr_version() >= 100


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", "/nonexistent/libdeepctr_hip.so")
    with pytest.raises(errors.NotFoundError):
        capi.lib()


def test_parser_matches_oracle_on_criteo_shaped_text():
    ids, vals, labels = O.synth_batch(257, 39, 117581, seed=3)
    text = O.to_libsvm(ids, vals, labels)
    pi, pv, pl = parse_libsvm(text, 39)
    oi, ov, ol = O.parse_libsvm(text, 39)
    assert np.array_equal(pi, oi) and np.array_equal(pi, ids)
    assert np.array_equal(pv.view(np.uint32), ov.view(np.uint32))        # bit-exact floats
    assert np.array_equal(pl, ol)


def test_parser_reference_comment_line_and_serving_sample():
    # the libsvm line in the comment at DeepFM.py:62 has 27 tokens (not 39): parses with field_size=27, ragged for 39
    line = ("1 1:0.5 2:0.03519 3:1 4:0.02567 7:0.03708 8:0.01705 9:0.06296 10:0.18185 11:0.02497 12:1 14:0.02565 15:0.03267 "
            "17:0.0247 18:0.03158 20:1 22:1 23:0.13169 24:0.02933 27:0.18159 31:0.0177 34:0.02888 38:1 51:1 63:1 132:1 164:1 236:1\n")
    i, v, l = parse_libsvm(line, 27)
    assert i.shape == (1, 27) and i[0, -1] == 236 and l[0] == 1.0 and v[0, 1] == np.float32(0.03519)
    with pytest.raises(errors.InvalidArgumentError):
        parse_libsvm(line, 39)
    s = "0 " + " ".join("%d:%s" % (a, repr(b)) for a, b in zip(SERVING_IDS, SERVING_VALS)) + "\n"
    i, v, l = parse_libsvm(s, 39)
    assert list(i[0]) == SERVING_IDS and np.array_equal(v[0], np.array(SERVING_VALS, np.float32))


@pytest.mark.parametrize("bad", ["1 1:0.5 2:\n", "1 1:0.5 x:2\n", "1 1:0.5 2:3:4\n", "abc 1:1 2:2\n", "1 1:0.5 2:1e\n", "1 3000000000:1 2:1\n"])
def test_parser_rejects_malformed_lines(bad):
    with pytest.raises(errors.InvalidArgumentError):
        parse_libsvm(bad, 2)


def test_parser_skips_empty_tokens_and_blank_lines():
    i, v, l = parse_libsvm("\n1  1:2   3:4 \n\n0 5:6 7:8\n", 2)
    assert i.tolist() == [[1, 3], [5, 7]] and v.tolist() == [[2.0, 4.0], [6.0, 8.0]] and l.tolist() == [1.0, 0.0]
    i, v, l = parse_libsvm("", 2)
    assert i.shape == (0, 2)


_float_text = st.one_of(
    st.floats(allow_nan=False, allow_infinity=False, width=64).map(lambda x: "%.17g" % x),
    st.floats(min_value=0, max_value=1).map(lambda x: ("%.6f" % x).rstrip("0").rstrip(".") or "0"),
    st.floats(allow_nan=False, allow_infinity=False, width=32).map(lambda x: "%.9g" % x),
    st.integers(min_value=0, max_value=10 ** 12).map(lambda n: "0.%012d" % n),
    st.tuples(st.integers(0, 10 ** 9), st.integers(0, 12)).map(lambda t: ("%d" % t[0])[:-t[1] or None] + "." + ("%d" % t[0])[-t[1]:] if t[1] else "%d" % t[0]),
)


@settings(max_examples=400, deadline=None)
@given(_float_text)
def test_float_parse_is_correctly_rounded_like_glibc_strtof(tok):
    libc = C.CDLL(None)
    libc.strtof.restype = C.c_float
    want = np.float32(libc.strtof(tok.encode(), None))
    _, v, _ = parse_libsvm("0 7:%s\n" % tok, 1)
    assert v[0, 0].view(np.uint32) == want.view(np.uint32) or (np.isnan(v[0, 0]) and np.isnan(want)), tok


def test_oracle_deepfm_matches_closed_form_and_fp64_shadow():
    cfg = O.Config(model="deepfm", field_size=39, feature_size=117581, embedding_size=8, deep_layers=(16, 8), dropout=(1, 1))
    p = O.init_params(cfg, seed=2, scale=0.05)
    ids = np.array([SERVING_IDS], dtype=np.int64)
    vals = np.array([SERVING_VALS], dtype=np.float32)
    out = O.forward(cfg, p, ids, vals)
    e = p["emb"].numpy()[ids[0]] * vals[0][:, None]
    yv = 0.5 * ((e.sum(0) ** 2) - (e ** 2).sum(0)).sum()
    pairs = sum(float(e[i] @ e[j]) for i in range(39) for j in range(i + 1, 39))      # FM identity: sum_{i<j} <e_i,e_j>
    assert abs(float(out["y_v"][0]) - yv) < 1e-6 and abs(yv - pairs) < 1e-5
    p64 = {k: v.double() for k, v in p.items()}
    out64 = O.forward(cfg, p64, ids, vals.astype(np.float64))
    assert abs(float(out["y"][0]) - float(out64["y"][0])) < 1e-5


def test_oracle_optimizers_one_step_closed_form():
    for kind, expect in [("Adam", lambda th, g, lr: th - lr * np.sign(g) * (1 / (1 + 1e-8 / np.abs(g)))),
                         ("Momentum", lambda th, g, lr: th - lr * g),
                         ("Adagrad", lambda th, g, lr: th - lr * g / np.sqrt(1e-8 + g * g))]:
        cfg = O.Config(model="fnn", field_size=2, feature_size=5, embedding_size=4, deep_layers=(4,), dropout=(1,), optimizer=kind,
                       learning_rate=0.1)
        p = O.init_params(cfg, seed=1, scale=0.1)
        th0 = p["mlp0/weights"].clone().numpy()
        opt = O.Optimizer(cfg, p)
        g = {k: torch.full_like(v, 0.25) for k, v in p.items()}
        opt.step(p, g)
        np.testing.assert_allclose(p["mlp0/weights"].numpy(), expect(th0, 0.25, 0.1), rtol=1e-5, atol=1e-7)


def test_oracle_streaming_auc():
    a = O.StreamingAUC()
    rng = np.random.default_rng(0)
    lab = (rng.random(5000) < 0.3).astype(np.float32)
    pred = np.clip(0.3 * lab + 0.7 * rng.random(5000), 0, 1).astype(np.float32)
    a.update(lab[:2000], pred[:2000])
    a.update(lab[2000:], pred[2000:])
    from sklearn.metrics import roc_auc_score
    assert abs(a.result() - roc_auc_score(lab, pred)) < 2e-3        # 200-threshold trapezoid vs exact AUC


def test_out_of_range_ids_raise_in_the_oracle():
    cfg = O.Config(model="deepfm", field_size=2, feature_size=5, embedding_size=4, deep_layers=(4,), dropout=(1,))
    p = O.init_params(cfg)
    with pytest.raises(IndexError):
        O.forward(cfg, p, np.array([[1, 5]]), np.ones((1, 2), np.float32))


def test_libsvm_dataset_batches_span_files_and_epochs(tmp_path):
    """repeat(epochs).batch(B) over several files (DeepFM.py:84-92): every batch but the last is full, order is preserved, the
    batch that spans a file / epoch edge is assembled from the carried tail and the next file's head."""
    from tf_repos_amd.input_pipeline import LibsvmDataset
    paths, parts = [], []
    for i, n in enumerate((10, 3, 25)):
        ids, vals, labels = O.synth_batch(n, 5, 100, seed=i)
        p = tmp_path / ("f%d.libsvm" % i)
        p.write_text(O.to_libsvm(ids, vals, labels))
        paths.append(str(p)); parts.append(ids)
    want = np.tile(np.concatenate(parts), (3, 1))
    for B in (4, 7, 16, 64, 200):
        got = list(LibsvmDataset(paths, 5, batch_size=B, num_epochs=3, binary_cache=False))
        assert np.array_equal(np.concatenate([g[0] for g in got]), want), B
        sizes = [len(g[2]) for g in got]
        assert all(s == B for s in sizes[:-1]) and 0 < sizes[-1] <= B, (B, sizes)


def test_streaming_dataset_yields_the_batches_of_the_whole_file_parse(tmp_path, monkeypatch):
    """LibsvmDataset(streaming=True): the file decoded chunk by chunk (chunk c + 1 in a background thread while chunk c is consumed),
    every epoch again -- the same batches, in the same order, as the whole-file parse, across chunk, file and epoch edges."""
    from tf_repos_amd.input_pipeline import LibsvmDataset
    F = 39
    paths, parts = [], []
    for i, n in enumerate((9000, 37, 6100)):             # ~3.5 MB, a few lines, ~2.4 MB of text: several 1-MB chunks per large file
        ids, vals, labels = O.synth_batch(n, F, 1_000_000, seed=20 + i)
        p = tmp_path / ("s%d.libsvm" % i)
        p.write_text(O.to_libsvm(ids, vals, labels))
        paths.append(str(p)); parts.append(ids)
    monkeypatch.setenv("DCTR_INPUT_CHUNK_MB", "1")
    for B in (256, 4096):
        whole = list(LibsvmDataset(paths, F, batch_size=B, num_epochs=2, binary_cache=False, streaming=False))
        stream = list(LibsvmDataset(paths, F, batch_size=B, num_epochs=2, streaming=True))
        assert len(whole) == len(stream)
        for a, b in zip(whole, stream):
            assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert not any(f.endswith(".npz") for f in os.listdir(tmp_path))          # streaming keeps nothing beside the file


def test_multithreaded_file_parse_matches_the_serial_parser(tmp_path):
    """dctr_parse_libsvm_mt (thread team inside the library): same rows as the serial decode for any thread count, blank lines
    and CRLF endings included; a malformed line reports the serial parser's message and line number."""
    from tf_repos_amd import errors
    from tf_repos_amd.input_pipeline import parse_file, parse_libsvm
    F = 39
    ids, vals, labels = O.synth_batch(3500, F, 1_000_000, seed=11)
    lines = O.to_libsvm(ids, vals, labels).split("\n")
    text = ""
    for r, ln in enumerate(lines):
        if not ln:
            continue
        text += ln + ("\r\n" if r % 7 == 0 else "\n")
        if r % 500 == 0:
            text += "   \n\n"
    assert len(text) > (1 << 20)                  # large enough for the library to use its threads
    p = tmp_path / "t.libsvm"
    p.write_text(text)
    ref = parse_libsvm(text, F)
    for threads in (1, 3, 8, 64):
        got = parse_file(str(p), F, threads=threads)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b), threads
    assert len(ref[2]) == 3500 and np.array_equal(ref[0], ids)
    bad = text.split("\n")
    bad[3000] = bad[3000].replace(":", ";", 1)
    p.write_text("\n".join(bad))
    with pytest.raises(errors.InvalidArgumentError, match="line 3001: token .* is not id:val"):
        parse_file(str(p), F, threads=8)


def test_multithreaded_csv_parse_matches_the_serial_decoder(tmp_path):
    """dctr_parse_csv_mt: the wide_n_deep record layout (1 float label + 13 float + 26 int columns, wide_n_deep.py:59-64), empty
    fields taking their defaults, decoded by any number of threads into the same arrays as one serial call."""
    from tf_repos_amd import errors
    from tf_repos_amd.input_pipeline import CsvDataset, parse_csv
    rng = np.random.default_rng(3)
    kinds = [0] * 14 + [1] * 26
    fd, idf = [0.0] * 14, [0] * 26
    lines = []
    for r in range(9000):
        f = ["%d" % rng.integers(0, 2)] + ["%.4f" % v if rng.random() > 0.1 else "" for v in rng.random(13)]
        i = ["%d" % v if rng.random() > 0.1 else "" for v in rng.integers(0, 10000, size=26)]
        lines.append(",".join(f + i))
        if r % 1000 == 0:
            lines.append("")
    text = "\n".join(lines) + "\n"
    assert len(text) > (1 << 20)
    p = tmp_path / "t.csv"
    p.write_text(text)
    rf, ri = parse_csv(text, kinds, fd, idf)
    assert len(rf) == 9000
    for threads in (1, 4, 32):
        f, i = CsvDataset([str(p)], kinds, fd, idf, threads=threads)._load(str(p))
        assert np.array_equal(f, rf) and np.array_equal(i, ri), threads
    lines[7000] = lines[7000].replace(",", ",x", 1)
    p.write_text("\n".join(lines) + "\n")
    with pytest.raises(errors.InvalidArgumentError, match="is not a valid float"):
        CsvDataset([str(p)], kinds, fd, idf, threads=8)._load(str(p))


def test_gemm_kernel_choice_per_shape():
    """Host logic of the layer products (csrc/gemm_dr.hip, no GPU): which kernel family and tile each shape of the BASELINE configs
    takes.  The direct-to-register kernel owns the one-round grids (c2 / c5 MLP), its 1x4 tile the small batches (c1, serving); grids
    of hundreds of rounds (AFM's 3 M pair rows) take the weights-stationary kernel, their long-reduction weight gradient the direct one."""
    import ctypes as C
    from tf_repos_amd import capi
    L = capi.lib()

    def plan(op, M, K, N):
        b = C.create_string_buffer(64)
        capi.check(L.dctr_gemm_plan(op.encode(), M, K, N, b, 64))
        return b.value.decode()
    assert [plan(o, 4096, 624, 400) for o in "fdw"] == ["dr 2x13", "dr 4x10", "dr 2x13 x6"]            # c2 layer 0
    assert [plan(o, 4096, 400, 400) for o in "fdw"] == ["dr 2x13", "dr 2x13", "dr 2x13 x9"]            # c2 layers 1, 2
    assert [plan(o, 256, 312, 400) for o in "fd"] == ["dr 1x4", "dr 1x4"]                              # c1 (README.md:49), B = 256
    assert plan("f", 1, 312, 400) == "dr 1x4"                                                          # one serving example
    assert [plan(o, 8192, 1989, 256) for o in "fdw"] == ["dr 4x8", "lds", "dr 2x16 x4"]                 # c4 inner-PNN layer 0: one round of 64 x 128 tiles forward
    assert [plan(o, 4096 * 741, 256, 256) for o in "fdw"] == ["ws", "ws", "dr 4x8 x32"]                # AFM.py:44,52 attention layer (64 x 128 tiles: 1.5x the flops per operand byte of 32 x 256)
    assert plan("w", 4096 * 741, 16, 256).startswith("lds")                                            # (K = 16: the fused AFM path anyway)
    with pytest.raises(Exception):
        plan("x", 1, 1, 1)


def test_dropout_mask_is_the_documented_function():
    """dctr_dropout_mask (host side of the engine's counter-based dropout, include/deepctr_hip.h "dropout sites") against a
    Python restatement of the documented formula; nn.dropout keeps with probability keep_prob (DeepFM.py:161-162)."""
    import ctypes as C
    from tf_repos_amd import capi
    L = capi.lib()
    M64 = (1 << 64) - 1

    def hash32(x):
        x ^= x >> 33; x = (x * 0xff51afd7ed558ccd) & M64; x ^= x >> 33; x = (x * 0xc4ceb9fe1a85ec53) & M64; x ^= x >> 33
        return x & 0xffffffff

    def ref(seed, t, site, n, keep):
        s = (seed ^ ((t * 0xD1B54A32D192ED03) & M64)) ^ site
        out = np.empty(n, np.uint8)
        for i in range(n):
            u = np.float32(hash32(s ^ ((i * 0x9E3779B97F4A7C15) & M64)) >> 8) * np.float32(1.0 / 16777216.0)
            out[i] = u >= np.float32(1.0) - np.float32(keep)
        return out

    def mask(seed, t, site, n, keep):
        m = np.empty(n, np.uint8)
        capi.check(L.dctr_dropout_mask(seed, t, site, n, keep, capi.ptr(m)))
        return m
    for seed, t, site, keep in [(0, 1, capi.SITE_MLP(0), 0.5), (1234, 7, capi.SITE_MLP2(2), 0.8), (2**63 + 5, 10**6, capi.SITE_AFM_ATT, 0.3)]:
        np.testing.assert_array_equal(mask(seed, t, site, 4000, keep), ref(seed, t, site, 4000, keep))
    big = mask(3, 2, capi.SITE_NFM_BI, 1 << 20, 0.8)
    assert abs(big.mean() - 0.8) < 2e-3                                        # kept with probability keep_prob
    assert mask(3, 2, capi.SITE_NFM_BI, 1000, 1.0).all()
    a = mask(3, 2, capi.SITE_MLP(0), 4096, 0.5)
    assert not np.array_equal(a, mask(3, 3, capi.SITE_MLP(0), 4096, 0.5))      # a new draw per step,
    assert not np.array_equal(a, mask(3, 2, capi.SITE_MLP(1), 4096, 0.5))      # per site
    assert not np.array_equal(a, mask(4, 2, capi.SITE_MLP(0), 4096, 0.5))      # and per seed
    assert abs(np.corrcoef(a[:-1], a[1:])[0, 1]) < 0.05                        # neighbours independent
    with pytest.raises(Exception):
        mask(0, 1, 0, 4, 0.0)


def test_round_robin_pair_schedule_covers_every_pair_once():
    """afm_pair_bwd_rr_kernel (csrc/afm.hip rr_pair) walks the F (F - 1) / 2 pairs of an example in round-robin-tournament order so
    that no field is touched twice inside a round (plain LDS read-modify-write, no atomics).  The same arithmetic here: every pair
    exactly once, rounds field-disjoint, for every field count the kernel can meet."""
    def rr_pair(n, F, r, m, half):
        a, c = n - 1, r
        if m > 0:
            a = r + m
            if a >= n - 1:
                a -= n - 1
            c = r - m
            if c < 0:
                c += n - 1
        lo, hi = (a, c) if a < c else (c, a)
        return (m < half and hi < F), lo, hi

    for F in range(2, 70):
        n = (F + 1) & ~1
        half = n // 2
        seen = set()
        for r in range(n - 1):
            fields = set()
            for m in range(half):
                ok, lo, hi = rr_pair(n, F, r, m, half)
                if not ok:
                    continue
                assert lo < hi < F and (lo, hi) not in seen
                assert lo not in fields and hi not in fields           # the round's pairs share no field
                seen.add((lo, hi))
                fields.update((lo, hi))
        assert len(seen) == F * (F - 1) // 2


def test_knob_list_names_every_knob_the_sources_read():
    """tools/KNOBS.txt (python tools/list_knobs.py > tools/KNOBS.txt) is the documentation of the DCTR_* A/B switches: a knob added to
    the sources and not to the list is a setting nobody can find.  Names only -- the line numbers in the file move with every edit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "list_knobs.py")], capture_output=True, text=True, check=True).stdout

    def names(text):
        return {ln.strip() for ln in text.split("\n") if ln.startswith("DCTR_")}
    listed = names(open(os.path.join(root, "tools", "KNOBS.txt")).read())
    assert names(out) == listed, sorted(names(out) ^ listed)
