"""Synthetic Criteo-shaped batches: the same generator feeds bench.py, the tests and the CPU baseline,
so every leg sees identical inputs.  Shapes follow get_criteo_feature.py:138-145: fields 1..13 numeric (id = field
index, value in [0,1) printed with 6 decimals), fields 14..39 categorical (field-disjoint id ranges, value 1)."""
import numpy as np


def synth_batch(B: int, F: int, V: int, seed: int, zipf: float = 1.05, uniform_ids: bool = False):
    rng = np.random.default_rng(seed)
    n_num = min(13, F)
    n_cat = F - n_num
    ids = np.zeros((B, F), dtype=np.int32)
    vals = np.ones((B, F), dtype=np.float32)
    for f in range(n_num):
        ids[:, f] = f + 1
        vals[:, f] = np.round(rng.random(B), 6).astype(np.float32)
    if n_cat > 0:
        share = max(1, (V - n_num) // n_cat)
        for c in range(n_cat):
            off = n_num + c * share
            r = np.minimum(rng.zipf(zipf, size=B) - 1, share - 1)
            ids[:, n_num + c] = np.minimum(off + r, V - 1)
    if uniform_ids:   # worst case for caches, best case for contention
        ids = rng.integers(0, V, size=(B, F)).astype(np.int32)
    labels = (rng.random(B) < 0.25).astype(np.float32)
    return ids, vals, labels


def to_libsvm(ids, vals, labels) -> str:
    """`label id:val ...` lines formatted like get_criteo_feature.py:141-148 ("%.6f" with trailing zeros stripped)."""
    lines = []
    for i in range(ids.shape[0]):
        toks = ["%d" % int(labels[i])]
        for f in range(ids.shape[1]):
            toks.append("%d:%s" % (ids[i, f], ("%.6f" % vals[i, f]).rstrip("0").rstrip(".") or "0"))
        lines.append(" ".join(toks))
    return "\n".join(lines) + "\n"
