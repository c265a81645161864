"""TFRecord files of tf.train.Example for the DIN / ESMM scripts (DIN.py:57-97, DeepCvrMTL.py:61-104).

Reading is the C parser (dctr_tfrecord_scan + dctr_examples_to_slot_csr): records go straight to the slot-ordered CSR that
dctr_train_step_csr consumes -- nothing is interpreted in Python per example.  Writing (TFRecordWriter / encode_example:
tf.python_io.TFRecordWriter + tf.train.Example.SerializeToString of Feature_pipeline/get_tfrecord.py:44-98) is a small
protobuf wire-format encoder; the framing checksums come from the C library."""
from __future__ import annotations

import ctypes as C
import struct
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np

from . import capi, errors


# ---- writing ---------------------------------------------------------------------------------------------------------------------
def _varint(v: int) -> bytes:
    v &= (1 << 64) - 1                      # int64 values are two's complement varints
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(field: int, payload: bytes) -> bytes:
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def encode_feature(kind: str, values: Sequence) -> bytes:
    """Feature{bytes_list=1 | float_list=2 | int64_list=3}, packed repeated values (what TF's python protobuf emits)."""
    if kind == "int64":
        return _ld(3, _ld(1, b"".join(_varint(int(v)) for v in values)) if len(values) else b"")
    if kind == "float":
        return _ld(2, _ld(1, struct.pack("<%df" % len(values), *[float(v) for v in values])) if len(values) else b"")
    if kind == "bytes":
        return _ld(1, b"".join(_ld(1, bytes(v)) for v in values))
    raise errors.InvalidArgumentError("unknown feature kind %r" % kind)


def encode_example(features: Dict[str, Tuple[str, Sequence]]) -> bytes:
    """Example{features=1: Features{feature=1: map<string, Feature>}}; keys in sorted order (deterministic serialisation)."""
    entries = b"".join(_ld(1, _ld(1, k.encode()) + _ld(2, encode_feature(kind, vals))) for k, (kind, vals) in sorted(features.items()))
    return _ld(1, entries)


# tf.train.{Example, Features, Feature, Int64List, FloatList, BytesList}: the message classes get_tfrecord.py:52-95 builds
class Int64List:
    kind = "int64"

    def __init__(self, value=()):
        self.value = [int(v) for v in value]


class FloatList:
    kind = "float"

    def __init__(self, value=()):
        self.value = [float(v) for v in value]


class BytesList:
    kind = "bytes"

    def __init__(self, value=()):
        self.value = [bytes(v) for v in value]


class Feature:
    def __init__(self, bytes_list=None, float_list=None, int64_list=None):
        given = [l for l in (bytes_list, float_list, int64_list) if l is not None]
        if len(given) > 1:
            raise errors.InvalidArgumentError("tf.train.Feature holds exactly one of bytes_list / float_list / int64_list")
        self.list = given[0] if given else None


class Features:
    def __init__(self, feature=None):
        self.feature = dict(feature or {})


class Example:
    def __init__(self, features=None):
        self.features = features or Features()

    def SerializeToString(self) -> bytes:
        return encode_example({k: (f.list.kind, f.list.value) if f.list is not None else ("bytes", []) for k, f in self.features.feature.items()})


class TFRecordWriter:
    """tf.python_io.TFRecordWriter(path): write(serialized_example) / close()  (get_tfrecord.py:47,96,101)."""

    def __init__(self, path: str):
        self._f = open(path, "wb")
        self._lib = capi.lib()

    def write(self, record: bytes) -> None:
        out = (C.c_uint8 * (len(record) + 16))()
        capi.check(self._lib.dctr_tfrecord_frame(record, len(record), out))
        self._f.write(bytes(out))

    def close(self) -> None:
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


# ---- reading ---------------------------------------------------------------------------------------------------------------------
class SlotSpec:
    """One parsed feature and where it lands in the MLP input (include/deepctr_hip.h dctr_slot_spec)."""

    def __init__(self, ids_feature: str, vals_feature: Optional[str] = None, fixed_len: int = -1):
        self.ids_feature, self.vals_feature, self.fixed_len = ids_feature, vals_feature, int(fixed_len)

    @property
    def n_slots(self) -> int:
        return self.fixed_len if self.fixed_len > 0 else 1

    def __repr__(self):
        return "SlotSpec(%r, %r, %d)" % (self.ids_feature, self.vals_feature, self.fixed_len)


def scan(buf: bytes, verify_crc: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    """(payload offsets, payload lengths) of the whole records in buf."""
    lib = capi.lib()
    n = C.c_int64()
    used = C.c_size_t()
    capi.check(lib.dctr_tfrecord_scan(buf, len(buf), -1, int(verify_crc), None, None, C.byref(n), C.byref(used)))
    if used.value != len(buf):          # TFRecordDataset raises DataLossError on a cut-off file; fewer examples is not an answer
        raise errors.DataLossError("truncated record at %d: the file holds %d bytes behind its last whole record" % (used.value, len(buf) - used.value))
    off = np.empty(n.value, np.int64)
    ln = np.empty(n.value, np.int64)
    capi.check(lib.dctr_tfrecord_scan(buf, len(buf), n.value, 0, capi.ptr(off), capi.ptr(ln), C.byref(n), None))
    return off, ln


def parse_slot_csr(buf: bytes, specs: Sequence[SlotSpec], label_names: Sequence[str], feature_size: int = 0, verify_crc: bool = True):
    """-> offsets int32 [N*S+1], ids int32 [nnz], weights f32 [nnz], labels f32 [n_labels, N]."""
    lib = capi.lib()
    off, ln = scan(buf, verify_crc)
    N = len(off)
    S = sum(s.n_slots for s in specs)
    arr = (capi.SlotSpec * len(specs))()
    keep = []
    for i, s in enumerate(specs):
        a = s.ids_feature.encode()
        b = s.vals_feature.encode() if s.vals_feature else None
        keep += [a, b]
        arr[i].ids_feature, arr[i].vals_feature, arr[i].fixed_len = a, b, s.fixed_len
    names = (C.c_char_p * max(len(label_names), 1))(*[n.encode() for n in label_names])
    n_ent = C.c_int64()
    args = (buf, capi.ptr(off), capi.ptr(ln), N, arr, len(specs), names, len(label_names), int(feature_size))
    capi.check(lib.dctr_examples_to_slot_csr(*args, 0, None, None, None, None, C.byref(n_ent)))
    offsets = np.empty(N * S + 1, np.int32)
    ids = np.empty(max(n_ent.value, 1), np.int32)
    wts = np.empty(max(n_ent.value, 1), np.float32)
    labels = np.empty((len(label_names), N), np.float32)
    capi.check(lib.dctr_examples_to_slot_csr(*args, n_ent.value, capi.ptr(offsets), capi.ptr(ids), capi.ptr(wts), capi.ptr(labels),
                                             C.byref(n_ent)))
    return offsets, ids[:n_ent.value], wts[:n_ent.value], labels


class TFRecordSlotDataset:
    """TFRecordDataset(files).map(parse).[shuffle].repeat(epochs).batch(B) (DIN.py:86-97) over the slot layout `specs`:
    yields (offsets int32 [b*S+1], ids int32 [nnz], weights f32 [nnz], labels f32 [n_labels, b]) numpy batches; the last
    batch of the stream may be short.  Files are parsed once (C parser) and kept as one CSR."""

    def __init__(self, filenames: Sequence[str], specs: Sequence[SlotSpec], label_names: Sequence[str], feature_size: int = 0,
                 batch_size: int = 32, num_epochs: int = 1, perform_shuffle: bool = False, seed: int = 0, verify_crc: bool = True):
        self.filenames = [filenames] if isinstance(filenames, str) else list(filenames)
        self.specs, self.label_names, self.feature_size = list(specs), list(label_names), int(feature_size)
        self.batch_size, self.num_epochs, self.perform_shuffle, self.seed = int(batch_size), int(num_epochs), perform_shuffle, seed
        self.verify_crc = verify_crc
        self.n_slots = sum(s.n_slots for s in self.specs)
        self._data = None

    def _load(self):
        if self._data is None:
            parts = []
            for path in self.filenames:
                with open(path, "rb") as f:
                    parts.append(parse_slot_csr(f.read(), self.specs, self.label_names, self.feature_size, self.verify_crc))
            base = np.cumsum([0] + [len(p[1]) for p in parts])
            # global entry offsets in int64 (several files together can pass 2^31 entries; take() rebases every batch to int32)
            offsets = np.concatenate([parts[0][0][:1].astype(np.int64)] + [p[0][1:].astype(np.int64) + int(b) for p, b in zip(parts, base[:-1])])
            self._data = (offsets, np.concatenate([p[1] for p in parts]), np.concatenate([p[2] for p in parts]),
                          np.concatenate([p[3] for p in parts], axis=1))
        return self._data

    @property
    def num_examples(self) -> int:
        return self._load()[3].shape[1]

    def take(self, rows: np.ndarray):
        """The CSR batch of the given example indices, in that order."""
        offsets, ids, wts, labels = self._load()
        S = self.n_slots
        rows = np.asarray(rows, np.int64)
        if len(rows) and np.array_equal(rows, np.arange(rows[0], rows[0] + len(rows))):      # contiguous: two slices
            s0, s1 = int(rows[0]) * S, (int(rows[-1]) + 1) * S
            e0, e1 = int(offsets[s0]), int(offsets[s1])
            return (offsets[s0:s1 + 1] - e0).astype(np.int32), ids[e0:e1], wts[e0:e1], labels[:, rows[0]:rows[-1] + 1]
        seg = (rows[:, None] * S + np.arange(S)[None, :]).ravel()
        lens = (offsets[seg + 1] - offsets[seg]).astype(np.int64)
        new_off = np.concatenate([[0], np.cumsum(lens)])
        src = np.repeat(offsets[seg].astype(np.int64) - new_off[:-1], lens) + np.arange(new_off[-1])
        return new_off.astype(np.int32), ids[src], wts[src], labels[:, rows]

    def __iter__(self) -> Iterator[Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]]:
        n = self.num_examples if self.filenames else 0
        if n == 0:
            return
        rng = np.random.default_rng(self.seed)
        # the stream of example indices over all epochs, cut into batches (batches may span an epoch boundary, as
        # repeat().batch() does)
        carry = np.empty(0, np.int64)
        for _ in range(self.num_epochs):
            order = rng.permutation(n) if self.perform_shuffle else np.arange(n)
            stream = np.concatenate([carry, order])
            full = len(stream) // self.batch_size * self.batch_size
            for i in range(0, full, self.batch_size):
                yield self.take(stream[i:i + self.batch_size])
            carry = stream[full:]
        if len(carry):
            yield self.take(carry)
