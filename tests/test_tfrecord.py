"""TFRecord / tf.train.Example input of the DIN / ESMM scripts (host code, no GPU): the C parser against the records'
known content and against the oracle's slot-ordered CSR of the same batch."""
import struct

import numpy as np
import pytest

from oracle import multihot_oracle as M
from tf_repos_amd import errors, tfrecord as T

# the parse spec of DIN.py:59-76 in the concat order of DIN.py:199
SPECS = [T.SlotSpec("feat_ids", None, 5), T.SlotSpec("u_catids", "u_catvals"), T.SlotSpec("u_shopids", "u_shopvals"),
         T.SlotSpec("u_brandids", "u_brandvals"), T.SlotSpec("u_intids", "u_intvals"), T.SlotSpec("a_catids", None, 0),
         T.SlotSpec("a_shopids", None, 0), T.SlotSpec("a_brandids", None, 0), T.SlotSpec("a_intids", None, -1)]


def example_of(batch, b):
    """The Example get_tfrecord.py:52-93 would write for example b of an oracle batch."""
    f = {"y": ("float", [batch["y"][b]]), "z": ("float", [batch["z"][b]]), "feat_ids": ("int64", batch["feat_ids"][b])}
    for n in M.MULTI_W:
        off, ids, vals = batch[n]
        f[n + "ids"] = ("int64", ids[off[b]:off[b + 1]])
        f[n + "vals"] = ("float", vals[off[b]:off[b + 1]])
    for n in M.SINGLE:
        f[n + "ids"] = ("int64", [batch[n][b]])
    off, ids, _ = batch["a_int"]
    f["a_intids"] = ("int64", ids[off[b]:off[b + 1]])
    return T.encode_example(f)


def write_file(path, batch):
    with T.TFRecordWriter(str(path)) as w:
        for b in range(len(batch["y"])):
            w.write(example_of(batch, b))


def test_known_framing_checksum():
    # crc32c("123456789") = 0xE3069283 (the CRC-32C check value); masked = rotr15 + 0xa282ead8
    import ctypes as C
    from tf_repos_amd import capi
    out = (C.c_uint8 * (9 + 16))()
    capi.check(capi.lib().dctr_tfrecord_frame(b"123456789", 9, out))
    raw = bytes(out)
    assert struct.unpack("<Q", raw[:8])[0] == 9 and raw[12:21] == b"123456789"
    c = 0xE3069283
    assert struct.unpack("<I", raw[21:25])[0] == ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


def test_parse_matches_oracle_slot_csr(tmp_path):
    cfg = M.Config(field_size=5, feature_size=500)
    batch = M.synth_batch(cfg, 37, seed=3)
    write_file(tmp_path / "a.tfrecord", batch)
    off, ids, wts, labels = T.parse_slot_csr((tmp_path / "a.tfrecord").read_bytes(), SPECS, ["y", "z"], cfg.feature_size)
    ro, ri, rw = M.slot_csr(cfg, batch)
    np.testing.assert_array_equal(off, ro)
    np.testing.assert_array_equal(ids, ri)                 # ids bit-exact
    np.testing.assert_array_equal(wts, rw)                 # float32 weights round-trip exactly
    np.testing.assert_array_equal(labels[0], batch["y"])
    np.testing.assert_array_equal(labels[1], batch["z"])


def test_dataset_batches_epochs_shuffle_and_files(tmp_path):
    cfg = M.Config(field_size=5, feature_size=500)
    b1, b2 = M.synth_batch(cfg, 10, seed=1), M.synth_batch(cfg, 7, seed=2)
    write_file(tmp_path / "p1", b1)
    write_file(tmp_path / "p2", b2)
    ds = T.TFRecordSlotDataset([str(tmp_path / "p1"), str(tmp_path / "p2")], SPECS, ["y"], cfg.feature_size, batch_size=4, num_epochs=2)
    assert ds.num_examples == 17 and ds.n_slots == cfg.n_slots
    batches = list(ds)
    assert [b[3].shape[1] for b in batches] == [4] * 8 + [2]          # repeat(2).batch(4): 34 examples, batches span the epoch edge
    ys = np.concatenate([b[3][0] for b in batches])
    np.testing.assert_array_equal(ys, np.tile(np.concatenate([b1["y"], b2["y"]]), 2))
    # an arbitrary row selection equals the slices of the per-example CSRs
    off, ids, wts, _ = ds.take(np.array([12, 3, 3]))
    o2, i2, w2 = M.slot_csr(cfg, b2)
    o1, i1, w1 = M.slot_csr(cfg, b1)
    S = cfg.n_slots
    want = np.concatenate([i2[o2[2 * S]:o2[3 * S]], i1[o1[3 * S]:o1[4 * S]], i1[o1[3 * S]:o1[4 * S]]])
    np.testing.assert_array_equal(ids, want)
    assert off[0] == 0 and off[-1] == len(want) and len(off) == 3 * S + 1
    sh = T.TFRecordSlotDataset([str(tmp_path / "p1")], SPECS, ["y"], cfg.feature_size, batch_size=10, perform_shuffle=True, seed=5)
    (o, i, w, l), = list(sh)
    assert sorted(l[0].tolist()) == sorted(b1["y"].tolist()) and len(i) == len(i1)


def test_unpacked_lists_missing_features_and_errors(tmp_path):
    V = 100
    # hand-built Example: unpacked int64 / float lists, an unknown extra feature, a VarLen feature that is absent (-> empty slot)
    def ld(field, payload):
        return T._varint((field << 3) | 2) + T._varint(len(payload)) + payload
    ids_unpacked = b"".join(T._varint((1 << 3) | 0) + T._varint(v) for v in (7, 8, 9))
    vals_unpacked = b"".join(T._varint((1 << 3) | 5) + struct.pack("<f", v) for v in (0.5, 1.5, 2.5))
    ent = lambda k, feat: ld(1, ld(1, k.encode()) + ld(2, feat))
    ex = ld(1, ent("u_catids", ld(3, ids_unpacked)) + ent("u_catvals", ld(2, vals_unpacked)) + ent("extra", ld(1, ld(1, b"xyz")))
            + ent("y", T.encode_feature("float", [1.0])) + ent("a_catids", T.encode_feature("int64", [42])))
    specs = [T.SlotSpec("u_catids", "u_catvals"), T.SlotSpec("a_catids", None, 0), T.SlotSpec("a_intids", None, -1)]
    path = tmp_path / "x"
    with T.TFRecordWriter(str(path)) as w:
        w.write(ex)
    off, ids, wts, labels = T.parse_slot_csr(path.read_bytes(), specs, ["y"], V)
    assert off.tolist() == [0, 3, 4, 4] and ids.tolist() == [7, 8, 9, 42] and wts.tolist() == [0.5, 1.5, 2.5, 1.0] and labels[0, 0] == 1.0
    raw = bytearray(path.read_bytes())
    with pytest.raises(errors.InvalidArgumentError, match="required"):          # FixedLenFeature without a default
        T.parse_slot_csr(bytes(raw), specs, ["z"], V)
    with pytest.raises(errors.InvalidArgumentError, match="outside"):           # out-of-range id
        T.parse_slot_csr(bytes(raw), specs, ["y"], 40)
    with pytest.raises(errors.InvalidArgumentError, match="weights"):           # ids / weights of different lengths
        T.parse_slot_csr(bytes(raw), [T.SlotSpec("a_catids", "u_catvals", -1)], ["y"], V)
    raw[20] ^= 0x01                                                             # flip a payload bit
    with pytest.raises(errors.InvalidArgumentError, match="checksum"):
        T.parse_slot_csr(bytes(raw), specs, ["y"], V)
    # a cut-off file is a DataLossError, as under tf.data.TFRecordDataset -- not a shorter dataset
    good = path.read_bytes()
    with pytest.raises(errors.DataLossError, match="truncated"):
        T.scan(good + good[:10])
    assert len(T.scan(good + good)[0]) == 2
