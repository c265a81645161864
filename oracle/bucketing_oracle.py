"""Restatement of the reference's Criteo id assignment -- TEST INFRASTRUCTURE (see oracle/deepctr_oracle.py header).

PINNED: checked bit-for-bit in tests/test_bucketing.py against tests/golden/criteo_small/*, which were produced by
running the reference's own deep_ctr/Feature_pipeline/get_criteo_feature.py in the build container
(tests/golden/make_bucketing_golden.py).  Only the id/value assignment is restated here (what the gather indexes);
file writing and the tr/va split live in the product module and are compared to the same fixtures.
"""
import collections
import sys

CLIP = [20, 600, 100, 50, 64000, 500, 100, 50, 500, 10, 10, 10, 50]      # get_criteo_feature.py:24


def build(train_lines, cutoff):
    """-> (mins, maxs, dicts, offsets).  get_criteo_feature.py:38-51 (dictionary), :74-85 (min/max), :120-123 (offsets)."""
    mins, maxs = [sys.maxsize] * 13, [-sys.maxsize] * 13
    counts = [collections.defaultdict(int) for _ in range(26)]
    for line in train_lines:
        feats = line.rstrip("\n").split("\t")
        for i in range(13):
            v = feats[1 + i]
            if v != "":
                v = min(int(v), CLIP[i])
                mins[i], maxs[i] = min(mins[i], v), max(maxs[i], v)
        for c in range(26):
            if feats[14 + c] != "":
                counts[c][feats[14 + c]] += 1
    dicts = []
    for c in range(26):
        items = sorted([kv for kv in counts[c].items() if kv[1] >= cutoff], key=lambda x: (-x[1], x[0]))   # :47-48
        d = dict(zip([k for k, _ in items], range(1, len(items) + 1)))                                     # :49-50
        d["<unk>"] = 0                                                                                     # :51
        dicts.append(d)
    offsets = [13]
    for c in range(26):
        offsets.append(offsets[c] + len(dicts[c]))
    return mins, maxs, dicts, offsets


def encode(feats, mins, maxs, dicts, offsets, shift=0):
    """One raw line (already split) -> (ids[39], value strings[39]).  get_criteo_feature.py:136-145 (shift=1: test.txt)."""
    ids, vals = [], []
    for i in range(13):
        v = feats[1 + i - shift]
        x = 0.0 if v == "" else (float(v) - mins[i]) / (maxs[i] - mins[i])                                 # :87-91
        ids.append(i + 1)
        vals.append("{0:.6f}".format(x).rstrip("0").rstrip("."))                                            # :141
    for c in range(26):
        ids.append(dicts[c].get(feats[14 + c - shift], 0) + offsets[c])                                     # :144
        vals.append("1")
    return ids, vals
