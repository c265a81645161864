"""input_fn side of the hot path (DeepFM.py:63-98): libsvm text -> batched (feat_ids, feat_vals, labels).

The text decode (K1) is the C ABI's dctr_parse_libsvm -- host C++ inside libdeepctr_hip.so, re-entrant, so a file is
parsed by a thread team inside the library (dctr_parse_libsvm_mt) like tf.data's
map(decode_libsvm, num_parallel_calls=10) (DeepFM.py:84).  Batches are staged in pinned host memory and copied to
the GPU on a side stream so the training step never waits for the parser.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np

from . import capi, errors


def parse_libsvm(text, field_size: int, max_rows: Optional[int] = None):
    """One call into the C parser.  text: str or bytes of whole lines.  Returns (ids i32 [n,F], vals f32 [n,F], labels f32 [n])."""
    if isinstance(text, str):
        text = text.encode()
    n_lines = text.count(b"\n") + (0 if text.endswith(b"\n") or not text else 1)
    cap = n_lines if max_rows is None else min(n_lines, max_rows)
    ids = np.empty((max(cap, 1), field_size), dtype=np.int32)
    vals = np.empty((max(cap, 1), field_size), dtype=np.float32)
    labels = np.empty(max(cap, 1), dtype=np.float32)
    n = C.c_int64()
    used = C.c_size_t()
    capi.check(capi.lib().dctr_parse_libsvm(text, len(text), field_size, cap, capi.ptr(ids), capi.ptr(vals), capi.ptr(labels),
                                            C.byref(n), C.byref(used)))
    k = n.value
    return ids[:k], vals[:k], labels[:k]


def parse_file(path: str, field_size: int, threads: int = 10):
    """A whole libsvm file through dctr_parse_libsvm_mt: the thread team lives inside the library (count pass, then every chunk
    parsed straight into its rows of the result -- no Python per chunk, no concatenation)."""
    # the file is MAPPED, not read: f.read() is a serial 150 MB copy out of the page cache (30-50 ms, as long as the whole 32-thread
    # parse); mapped pages are first touched by the parse threads themselves
    # (the mapping is read live by both passes: the file must not change underneath them -- a dataset that is being rewritten is
    #  refused below instead of surfacing as a SIGBUS or as two passes that disagree)
    st0 = os.stat(path)
    size = st0.st_size
    buf = np.memmap(path, dtype=np.uint8, mode="r") if size > 0 else np.zeros(0, dtype=np.uint8)
    lib = capi.lib()
    n = C.c_int64()
    capi.check(lib.dctr_parse_libsvm_mt(capi.ptr(buf), size, field_size, int(threads), None, None, None, 0, C.byref(n)))
    rows = n.value
    ids = np.empty((max(rows, 1), field_size), dtype=np.int32)
    vals = np.empty((max(rows, 1), field_size), dtype=np.float32)
    labels = np.empty(max(rows, 1), dtype=np.float32)
    capi.check(lib.dctr_parse_libsvm_mt(capi.ptr(buf), size, field_size, int(threads), capi.ptr(ids), capi.ptr(vals), capi.ptr(labels), rows,
                                        C.byref(n)))
    del buf
    st1 = os.stat(path)
    if (st1.st_size, st1.st_mtime_ns) != (st0.st_size, st0.st_mtime_ns):
        raise errors.InvalidArgumentError("%s changed while it was being parsed (size %d -> %d): parse a file that is not being written" % (
            path, st0.st_size, st1.st_size))
    return ids[:rows], vals[:rows], labels[:rows]


def _line_chunks(buf, size: int, target: int):
    """(start, end) byte spans of whole lines, ~target bytes each"""
    pos = 0
    while pos < size:
        end = min(size, pos + target)
        while end < size:                       # extend to the end of the line the cut fell into
            win = np.asarray(buf[end:min(size, end + 65536)])
            nl = np.flatnonzero(win == 10)
            if nl.size:
                end += int(nl[0]) + 1
                break
            end = min(size, end + 65536)
        yield pos, end
        pos = end


def parse_span(buf, start: int, end: int, field_size: int, threads: int):
    """bytes [start, end) of a mapped libsvm file (whole lines) through the library's thread team"""
    lib = capi.lib()
    view = buf[start:end]
    n = C.c_int64()
    capi.check(lib.dctr_parse_libsvm_mt(capi.ptr(view), end - start, field_size, int(threads), None, None, None, 0, C.byref(n)))
    rows = n.value
    ids = np.empty((max(rows, 1), field_size), dtype=np.int32)
    vals = np.empty((max(rows, 1), field_size), dtype=np.float32)
    labels = np.empty(max(rows, 1), dtype=np.float32)
    capi.check(lib.dctr_parse_libsvm_mt(capi.ptr(view), end - start, field_size, int(threads), capi.ptr(ids), capi.ptr(vals), capi.ptr(labels), rows,
                                        C.byref(n)))
    return ids[:rows], vals[:rows], labels[:rows]


class LibsvmDataset:
    """TextLineDataset(filenames).map(decode_libsvm, 10).prefetch().[shuffle(256)].repeat(num_epochs).batch(batch_size)
    (DeepFM.py:84-92) as a Python iterator of numpy batches; the last batch may be short.

    Two ways through a file.  Default: the whole file is decoded once (thread team inside the library), kept in memory and in a
    `.dctr.npz` beside it -- later epochs and later calls replay the parsed rows.  `streaming=True` (DCTR_INPUT_STREAMING=1; what
    tf.data does): the file is decoded in chunks of DCTR_INPUT_CHUNK_MB (64) megabytes of whole lines, chunk c + 1 by a background
    thread while the batches of chunk c are consumed; nothing is kept, every epoch decodes the text again, memory stays at two chunks
    whatever the file's size."""

    def __init__(self, filenames: Sequence[str], field_size: int, batch_size: int = 32, num_epochs: int = 1,
                 perform_shuffle: bool = False, threads: int = 10, seed: int = 0, binary_cache: bool = True, streaming: Optional[bool] = None):
        self.filenames = [filenames] if isinstance(filenames, str) else list(filenames)
        self.field_size = field_size
        self.batch_size = batch_size
        self.num_epochs = num_epochs
        self.perform_shuffle = perform_shuffle
        self.threads = threads
        self.seed = seed
        self.binary_cache = binary_cache
        self.streaming = (os.environ.get("DCTR_INPUT_STREAMING", "0") == "1") if streaming is None else bool(streaming)
        self._cache = {}

    def _pieces(self, path):
        """the file's parsed rows, piece by piece: one piece (the whole file) or, streaming, one per chunk"""
        if not self.streaming:
            yield self._load(path)
            return
        from concurrent.futures import ThreadPoolExecutor
        st0 = os.stat(path)
        size = st0.st_size
        if size == 0:
            return
        buf = np.memmap(path, dtype=np.uint8, mode="r")
        target = max(1, int(os.environ.get("DCTR_INPUT_CHUNK_MB", "64"))) << 20
        spans = _line_chunks(buf, size, target)
        with ThreadPoolExecutor(1) as ex:           # the ctypes call releases the GIL: chunk c + 1 is decoded while chunk c is consumed
            nxt = next(spans, None)
            fut = ex.submit(parse_span, buf, nxt[0], nxt[1], self.field_size, self.threads) if nxt else None
            while fut is not None:
                data = fut.result()
                nxt = next(spans, None)
                fut = ex.submit(parse_span, buf, nxt[0], nxt[1], self.field_size, self.threads) if nxt else None
                yield data
        del buf
        st1 = os.stat(path)
        if (st1.st_size, st1.st_mtime_ns) != (st0.st_size, st0.st_mtime_ns):
            raise errors.InvalidArgumentError("%s changed while it was being parsed (size %d -> %d): parse a file that is not being written" % (
                path, st0.st_size, st1.st_size))

    def _load(self, path):
        if path in self._cache:
            return self._cache[path]
        npz = path + ".f%d.dctr.npz" % self.field_size
        st = os.stat(path)
        data = None
        if self.binary_cache and os.path.exists(npz):
            # the cache names the exact source it was made from (size + mtime_ns): a dataset replaced by an OLDER file (cp -p,
            # rsync) must not serve stale rows, and a cache another process is still writing (or died writing) is just re-parsed
            try:
                z = np.load(npz)
                if int(z["src_size"]) == st.st_size and int(z["src_mtime_ns"]) == st.st_mtime_ns:
                    data = (z["ids"], z["vals"], z["labels"])
            except Exception:           # noqa: BLE001  (BadZipFile, KeyError of an old-format cache, truncated file ...)
                data = None
        if data is None:
            data = parse_file(path, self.field_size, self.threads)
            if self.binary_cache:
                tmp = npz + ".tmp.%d.npz" % os.getpid()
                try:                    # pre-tokenised cache, written atomically: several ranks load the same file
                    np.savez(tmp, ids=data[0], vals=data[1], labels=data[2], src_size=np.int64(st.st_size), src_mtime_ns=np.int64(st.st_mtime_ns))
                    os.replace(tmp, npz)
                except OSError:
                    try:
                        os.remove(tmp)
                    except OSError:
                        pass
        self._cache[path] = data
        return data

    def __iter__(self) -> Iterator[Tuple[np.ndarray, np.ndarray, np.ndarray]]:
        rng = np.random.default_rng(self.seed)
        B = self.batch_size
        carry = None
        for _epoch in range(self.num_epochs):
            for path in self.filenames:
              for ids, vals, labels in self._pieces(path):
                if self.perform_shuffle:          # shuffle(buffer_size=256): windowed shuffle (DeepFM.py:88)
                    n = len(labels)
                    perm = np.arange(n)
                    for s in range(0, n, 256):
                        rng.shuffle(perm[s:s + 256])
                    ids, vals, labels = ids[perm], vals[perm], labels[perm]
                n = len(labels)
                s0 = 0
                if carry is not None:
                    # repeat().batch(): the batch that spans the file / epoch edge = the carried tail + the head of this file
                    # (only that one batch is assembled by copy; the rest are views)
                    take = min(B - len(carry[2]), n)
                    carry = (np.concatenate([carry[0], ids[:take]]), np.concatenate([carry[1], vals[:take]]),
                             np.concatenate([carry[2], labels[:take]]))
                    s0 = take
                    if len(carry[2]) == B:
                        yield carry
                        carry = None
                    else:
                        continue                # this file was shorter than the gap: keep filling from the next one
                full = s0 + (n - s0) // B * B
                for s in range(s0, full, B):
                    yield ids[s:s + B], vals[s:s + B], labels[s:s + B]
                if full < n:
                    carry = (ids[full:], vals[full:], labels[full:])
        if carry is not None and len(carry[2]) > 0:
            yield carry


def parse_csv(text, kinds: Sequence[int], f_defaults: Sequence[float], i_defaults: Sequence[int], max_rows: Optional[int] = None):
    """One call into the C CSV decoder (dctr_parse_csv = tf.decode_csv, wide_n_deep.py:67-73).  kinds[c]: 0 float / 1 int32.
    Returns (floats f32 [n, n_float], ints i32 [n, n_int]) with the columns of each kind in order of appearance."""
    if isinstance(text, str):
        text = text.encode()
    n_lines = text.count(b"\n") + (0 if text.endswith(b"\n") or not text else 1)
    cap = n_lines if max_rows is None else min(n_lines, max_rows)
    k = np.asarray(kinds, dtype=np.int8)
    nf, ni = int((k == 0).sum()), int((k == 1).sum())
    fd = np.asarray(list(f_defaults) or [0.0], dtype=np.float32)
    idf = np.asarray(list(i_defaults) or [0], dtype=np.int32)
    out_f = np.empty((max(cap, 1), max(nf, 1)), dtype=np.float32)
    out_i = np.empty((max(cap, 1), max(ni, 1)), dtype=np.int32)
    n = C.c_int64()
    used = C.c_size_t()
    capi.check(capi.lib().dctr_parse_csv(text, len(text), len(k), capi.ptr(k), capi.ptr(fd), capi.ptr(idf), cap, capi.ptr(out_f),
                                         capi.ptr(out_i), C.byref(n), C.byref(used)))
    return out_f[:n.value, :nf], out_i[:n.value, :ni]


class CsvDataset:
    """TextLineDataset(filenames).map(parse_csv, 10).prefetch().repeat(num_epochs).batch(batch_size) (wide_n_deep.py:66-89) as
    an iterator of numpy batches (floats [b, n_float], ints [b, n_int]); files are decoded by the library's thread team
    (dctr_parse_csv_mt)."""

    def __init__(self, filenames: Sequence[str], kinds: Sequence[int], f_defaults: Sequence[float], i_defaults: Sequence[int],
                 batch_size: int = 1, num_epochs: int = 1, threads: int = 10):
        self.filenames = [filenames] if isinstance(filenames, str) else list(filenames)
        self.kinds, self.f_defaults, self.i_defaults = list(kinds), list(f_defaults), list(i_defaults)
        self.batch_size, self.num_epochs, self.threads = batch_size, num_epochs, threads
        self._cache = {}

    def _load(self, path):
        if path not in self._cache:
            size = os.path.getsize(path)           # (mapped, not read: see parse_file)
            buf = np.memmap(path, dtype=np.uint8, mode="r") if size > 0 else np.zeros(0, dtype=np.uint8)
            lib = capi.lib()
            k = np.asarray(self.kinds, dtype=np.int8)
            nf, ni = int((k == 0).sum()), int((k == 1).sum())
            fd = np.asarray(self.f_defaults, dtype=np.float32) if nf else np.zeros(1, np.float32)
            idf = np.asarray(self.i_defaults, dtype=np.int32) if ni else np.zeros(1, np.int32)
            n = C.c_int64()
            args = (capi.ptr(buf), size, len(k), capi.ptr(k), capi.ptr(fd), capi.ptr(idf), int(self.threads))
            capi.check(lib.dctr_parse_csv_mt(*args, None, None, 0, C.byref(n)))         # count, then parse in place
            rows = n.value
            out_f = np.empty((max(rows, 1), max(nf, 1)), dtype=np.float32)
            out_i = np.empty((max(rows, 1), max(ni, 1)), dtype=np.int32)
            capi.check(lib.dctr_parse_csv_mt(*args, capi.ptr(out_f), capi.ptr(out_i), rows, C.byref(n)))
            self._cache[path] = (out_f[:rows, :nf], out_i[:rows, :ni])
        return self._cache[path]

    def __iter__(self):
        B = self.batch_size
        carry = None
        for _epoch in range(self.num_epochs):
            for path in self.filenames:
                f, i = self._load(path)
                n = len(f)
                s0 = 0
                if carry is not None:       # only the batch that spans the file / epoch edge is assembled by copy
                    take = min(B - len(carry[0]), n)
                    carry = (np.concatenate([carry[0], f[:take]]), np.concatenate([carry[1], i[:take]]))
                    s0 = take
                    if len(carry[0]) == B:
                        yield carry
                        carry = None
                    else:
                        continue
                full = s0 + (n - s0) // B * B
                for s in range(s0, full, B):
                    yield f[s:s + B], i[s:s + B]
                if full < n:
                    carry = (f[full:], i[full:])
        if carry is not None and len(carry[0]) > 0:
            yield carry
