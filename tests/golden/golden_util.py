"""Shared by tests/golden/make_model_golden.py (writer) and tests/test_model_golden.py (reader)."""
import ast

import numpy as np

BIG = 20000


def draw_named(shapes, seed, scale):
    """Seeded N(0, scale) fp32 values for {variable name: shape} in sorted-name order; batch-norm statistics / scales at their
    TF initial values.  The fixtures store the seed, not the weights."""
    rng = np.random.default_rng(seed)
    out = {}
    for name in sorted(shapes):
        shp = tuple(shapes[name])
        if name.endswith("moving_variance") or name.endswith("gamma"):
            out[name] = np.ones(shp, np.float32)
        elif name.endswith("moving_mean") or name.endswith("beta"):
            out[name] = np.zeros(shp, np.float32)
        else:
            out[name] = rng.normal(0, scale, size=shp).astype(np.float32)
    return out


def store(out, key, arr):
    """Small tensors whole; of a big one (the 400-wide layers, Outer-PNN's first layer) 2048 seeded samples + two checksums."""
    arr = np.asarray(arr)
    if arr.size <= BIG:
        out[key] = arr
        return
    idx = np.random.default_rng(arr.size).choice(arr.size, 2048, replace=False)
    flat = arr.reshape(-1)
    out[key + "@idx"], out[key + "@val"] = idx.astype(np.int64), flat[idx]
    out[key + "@sum"], out[key + "@sq"] = np.float64(flat.sum()), np.float64(np.square(flat).sum())


def max_err(fx, key, got):
    """max |got - expected| over what the fixture holds of `key` (everything, or the samples + checksums scaled to elements)."""
    got = np.asarray(got, dtype=np.float64)
    if key in fx:
        return float(np.abs(got.reshape(fx[key].shape) - fx[key]).max())
    idx, val = fx[key + "@idx"], fx[key + "@val"]
    e = float(np.abs(got.reshape(-1)[idx] - val).max())
    e = max(e, abs(float(got.sum()) - float(fx[key + "@sum"])) / np.sqrt(got.size))
    return e


def meta(fx, key):
    v = fx[key]
    v = v.item() if hasattr(v, "item") else v
    return ast.literal_eval(v) if isinstance(v, str) and v[:1] in "[({" else v
