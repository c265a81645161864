"""TensorFlow checkpoint bundles ("V2" checkpoints: `<prefix>.index` + `<prefix>.data-00000-of-00001`), read and written
without TensorFlow -- the container behind `tf.estimator.Estimator(model_dir=...)` (DeepFM.py:288,341): a model trained by the
reference under TF can be loaded here by its variable names (fm_bias / fm_w / fm_v, Deep-part/mlp0/weights
...), and what this engine trains can be handed back.  SURVEY 8f row 2.

Format [TF tensor_bundle / LevelDB table format, restated from the published format, not from TF sources -- UNPINNED: no
TF-written checkpoint exists in the reference tree or in this container to test against; the tests round-trip and check the
structural invariants (magic, block checksums, restart arrays)]:
  index file = an SSTable: data blocks of prefix-compressed (key, value) entries + a restart array, each block followed by a
      1-byte compression tag (0 none, 1 snappy) and the masked crc32c of block + tag; a metaindex block, an index block whose
      values are BlockHandles (varint offset, varint size), and a 48-byte footer ending in the magic 0xdb4775248b80fb57.
  key ""   -> BundleHeaderProto {1: num_shards, 2: endianness (0 little), 3: VersionDef{1: producer}}
  key name -> BundleEntryProto  {1: dtype (1 f32, 3 i32, 9 i64), 2: TensorShapeProto{2: Dim{1: size}}, 3: shard_id, 4: offset,
                                 5: size, 6: masked crc32c of the bytes (fixed32)}
  data file = the tensors' little-endian row-major bytes at those offsets.
"""
from __future__ import annotations

import ctypes as C
import os
import struct
from typing import Dict, List, Tuple

import numpy as np

from . import capi, errors

MAGIC = 0xdb4775248b80fb57
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64}
DTYPE_OF = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.int64): 9}


def masked_crc32c(buf: bytes) -> int:
    out = C.c_uint32()
    capi.check(capi.lib().dctr_crc32c(bytes(buf), len(buf), 1, C.byref(out)))
    return int(out.value)


# ---- varints / protobuf wire format ---------------------------------------------------------------------------------------
def _varint(buf, pos) -> Tuple[int, int]:
    r = s = 0
    while True:
        b = buf[pos]
        pos += 1
        r |= (b & 0x7F) << s
        if not b & 0x80:
            return r, pos
        s += 7


def _enc_varint(v: int) -> bytes:
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _pb_fields(buf: bytes):
    pos = 0
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise errors.DataLossError("checkpoint index: unsupported protobuf wire type %d" % wt)
        yield f, wt, v


def _pb(field: int, wt: int, v) -> bytes:
    key = _enc_varint(field << 3 | wt)
    if wt == 0:
        return key + _enc_varint(int(v))
    if wt == 2:
        return key + _enc_varint(len(v)) + bytes(v)
    if wt == 5:
        return key + struct.pack("<I", int(v))
    raise ValueError(wt)


def _snappy_decompress(buf: bytes) -> bytes:
    n, pos = _varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        t = tag & 3
        if t == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if t == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif t == 2:
            ln = (tag >> 2) + 1
            off = buf[pos] | buf[pos + 1] << 8
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        for _ in range(ln):
            out.append(out[-off])
    if len(out) != n:
        raise errors.DataLossError("checkpoint index: corrupt snappy block")
    return bytes(out)


# ---- SSTable --------------------------------------------------------------------------------------------------------------
def _read_block(buf: bytes, off: int, size: int, verify: bool) -> bytes:
    raw, tag = buf[off:off + size], buf[off + size]
    if verify:
        crc = struct.unpack_from("<I", buf, off + size + 1)[0]
        if crc != masked_crc32c(buf[off:off + size + 1]):
            raise errors.DataLossError("checkpoint index: block checksum mismatch at offset %d" % off)
    if tag == 0:
        return bytes(raw)
    if tag == 1:
        return _snappy_decompress(bytes(raw))
    raise errors.DataLossError("checkpoint index: unknown block compression %d" % tag)


def _block_entries(block: bytes) -> List[Tuple[bytes, bytes]]:
    nrest = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * nrest
    out, pos, key = [], 0, b""
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        out.append((key, block[pos:pos + vlen]))
        pos += vlen
    return out


def read_index(path: str, verify: bool = True) -> Dict[str, Dict]:
    buf = open(path, "rb").read()
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != MAGIC:
        raise errors.DataLossError("%s is not a TensorFlow checkpoint index (bad table magic)" % path)
    foot = buf[-48:]
    _mo, p = _varint(foot, 0)
    _ms, p = _varint(foot, p)
    io, p = _varint(foot, p)
    isz, p = _varint(foot, p)
    entries = {}
    for _sep, handle in _block_entries(_read_block(buf, io, isz, verify)):
        bo, q = _varint(handle, 0)
        bs, q = _varint(handle, q)
        for key, val in _block_entries(_read_block(buf, bo, bs, verify)):
            e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None}
            if key == b"":
                hdr = {f: v for f, _w, v in _pb_fields(val)}
                if hdr.get(2, 0) != 0:
                    raise errors.UnimplementedError("big-endian checkpoint bundles")
                entries[""] = {"num_shards": hdr.get(1, 1)}
                continue
            for f, _w, v in _pb_fields(val):
                if f == 1:
                    e["dtype"] = v
                elif f == 2:
                    for f2, _w2, dim in _pb_fields(v):
                        if f2 == 2:
                            d = {ff: vv for ff, _ww, vv in _pb_fields(dim)}
                            e["shape"].append(int(d.get(1, 0)))
                elif f == 3:
                    e["shard_id"] = v
                elif f == 4:
                    e["offset"] = v
                elif f == 5:
                    e["size"] = v
                elif f == 6:
                    e["crc32c"] = v
                elif f == 7:
                    raise errors.UnimplementedError("partitioned (sliced) variables in a checkpoint bundle: %s" % key.decode())
            entries[key.decode()] = e
    return entries


def read_bundle(prefix: str, verify: bool = True) -> Dict[str, np.ndarray]:
    """{variable name: array} of `<prefix>.index` / `<prefix>.data-0000i-of-0000n`."""
    idx = read_index(prefix + ".index", verify)
    nsh = idx.pop("", {"num_shards": 1})["num_shards"]
    shards = {}
    out = {}
    for name, e in idx.items():
        if e["dtype"] not in DTYPES:
            raise errors.UnimplementedError("checkpoint tensor %s has dtype %d (only float32/float64/int32/int64 are read)" % (name, e["dtype"]))
        sid = e["shard_id"]
        if sid not in shards:
            shards[sid] = np.memmap("%s.data-%05d-of-%05d" % (prefix, sid, nsh), dtype=np.uint8, mode="r")
        raw = shards[sid][e["offset"]:e["offset"] + e["size"]]
        if verify and e["crc32c"] is not None and masked_crc32c(raw.tobytes()) != e["crc32c"]:
            raise errors.DataLossError("checkpoint tensor %s: checksum mismatch" % name)
        out[name] = np.frombuffer(raw.tobytes(), dtype=DTYPES[e["dtype"]]).reshape(e["shape"])
    return out


def _build_block(items: List[Tuple[bytes, bytes]], restart_interval: int = 16) -> bytes:
    out, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += _enc_varint(shared) + _enc_varint(len(k) - shared) + _enc_varint(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_bundle(prefix: str, tensors: Dict[str, np.ndarray], block_size: int = 4096) -> None:
    """One shard, uncompressed blocks (what TF's BundleWriter emits), keys in sorted order."""
    os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
    items: List[Tuple[bytes, bytes]] = [(b"", _pb(1, 0, 1) + _pb(2, 0, 0) + _pb(3, 2, _pb(1, 0, 1)))]
    off = 0
    with open(prefix + ".data-00000-of-00001.tmp", "wb") as f:
        for name in sorted(tensors):
            a = np.asarray(tensors[name])
            a = a if a.flags.c_contiguous else np.ascontiguousarray(a)        # (ascontiguousarray would turn a scalar into shape [1])
            if a.dtype not in DTYPE_OF:
                raise errors.InvalidArgumentError("cannot write dtype %s" % a.dtype)
            raw = a.tobytes()
            shape = b"".join(_pb(2, 2, _pb(1, 0, d)) for d in a.shape)
            ent = _pb(1, 0, DTYPE_OF[a.dtype]) + _pb(2, 2, shape) + (_pb(4, 0, off) if off else b"") + _pb(5, 0, len(raw)) + _pb(6, 5, masked_crc32c(raw))
            items.append((name.encode(), ent))
            f.write(raw)
            off += len(raw)
    out = bytearray()
    index_items = []

    def emit(block_items):
        blk = _build_block(block_items) + b"\x00"
        handle = _enc_varint(len(out)) + _enc_varint(len(blk) - 1)
        out.extend(blk + struct.pack("<I", masked_crc32c(blk)))
        return handle
    cur, cur_bytes = [], 0
    for k, v in items:
        cur.append((k, v))
        cur_bytes += len(k) + len(v) + 6
        if cur_bytes >= block_size:
            index_items.append((cur[-1][0], emit(cur)))
            cur, cur_bytes = [], 0
    if cur:
        index_items.append((cur[-1][0], emit(cur)))
    meta_handle = emit([])
    index_handle = emit(index_items)          # (separator keys = the last key of each block: a valid, if not shortest, choice)
    foot = meta_handle + index_handle
    out += foot + b"\x00" * (40 - len(foot)) + struct.pack("<Q", MAGIC)
    with open(prefix + ".index.tmp", "wb") as f:
        f.write(bytes(out))
    os.replace(prefix + ".data-00000-of-00001.tmp", prefix + ".data-00000-of-00001")
    os.replace(prefix + ".index.tmp", prefix + ".index")


# ---- Estimator glue: TF's slot-variable names <-> the engine's two slot arrays ----------------------------------------------
SLOT_NAMES = {"Adam": ("Adam", "Adam_1"), "Adagrad": ("Adagrad", None), "Momentum": ("Momentum", None), "ftrl": ("Ftrl", "Ftrl_1")}


def latest_tf_checkpoint(model_dir: str):
    """`checkpoint` state file of tf.train.Saver (model_checkpoint_path: "model.ckpt-N"), else the highest-numbered .index."""
    state = os.path.join(model_dir, "checkpoint")
    if os.path.exists(state):
        for line in open(state):
            if line.startswith("model_checkpoint_path:"):
                p = line.split(":", 1)[1].strip().strip('"')
                p = p if os.path.isabs(p) else os.path.join(model_dir, p)
                if os.path.exists(p + ".index"):
                    return p
    best = None
    for f in os.listdir(model_dir) if os.path.isdir(model_dir) else []:
        if f.startswith("model.ckpt-") and f.endswith(".index"):
            try:
                n = int(f[len("model.ckpt-"):-len(".index")])
            except ValueError:
                continue
            if best is None or n > best[0]:
                best = (n, os.path.join(model_dir, f[:-len(".index")]))
    return best[1] if best else None


def bundle_to_state(tensors: Dict[str, np.ndarray], optimizer: str) -> Dict[str, np.ndarray]:
    """TF names -> the Estimator's snapshot keys (`name`, `name/slot0`, `name/slot1`, `global_step`)."""
    s0, s1 = SLOT_NAMES.get(optimizer, (None, None))
    out = {}
    for k, v in tensors.items():
        if k in ("beta1_power", "beta2_power"):      # Adam's non-slot variables: implied by global_step here
            continue
        base, _, last = k.rpartition("/")
        if s0 and last == s0 and base in tensors:
            out[base + "/slot0"] = v
        elif s1 and last == s1 and base in tensors:
            out[base + "/slot1"] = v
        else:
            out[k] = v
    return out


def state_to_bundle(state: Dict[str, np.ndarray], optimizer: str, beta1: float = 0.9, beta2: float = 0.999) -> Dict[str, np.ndarray]:
    s0, s1 = SLOT_NAMES.get(optimizer, (None, None))
    out = {}
    for k, v in state.items():
        if k.endswith("/slot0"):
            if s0:
                out[k[:-6] + "/" + s0] = np.asarray(v)
        elif k.endswith("/slot1"):
            if s1:
                out[k[:-6] + "/" + s1] = np.asarray(v)
        else:
            out[k] = np.asarray(v, dtype=np.int64) if k == "global_step" else np.asarray(v)
    if optimizer == "Adam" and "global_step" in state:
        # AdamOptimizer's two non-slot variables [TF-1.x]: a TRAINING graph's Saver looks for them (eval / predict graphs do not).
        # TF creates them at beta (not 1) and multiplies by beta AFTER each apply (`_finish`), reading the current value for
        # lr_t = lr sqrt(1 - beta2_power) / (1 - beta1_power): after global_step = t applies they hold beta^(t+1), which is what
        # the NEXT step (the engine's step t+1, bias-corrected with beta^(t+1)) needs.  beta^t here would make a resumed TF graph
        # one step off, and at t = 0 divide by zero.  The betas are the engine's (DeepFM.py:205 passes TF's defaults 0.9 / 0.999).
        t = int(np.asarray(state["global_step"]))
        out["beta1_power"] = np.float32(float(beta1) ** (t + 1))
        out["beta2_power"] = np.float32(float(beta2) ** (t + 1))
    return out
