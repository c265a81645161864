"""Criteo raw TSV -> tr/va/te.libsvm + feature_map: the integer-bucketing front of the hot path.

Drop-in for deep_ctr/Feature_pipeline/get_criteo_feature.py (same flags: --input_dir --output_dir --cutoff --threads,
same four output files, byte-identical contents -- checked against the reference's own output in
tests/test_bucketing.py).  Semantics kept bit-exact (get_criteo_feature.py:28-61, 64-91, 97-167):
  * categorical id = frequency rank among values seen >= cutoff times, ties broken by the value string
    (sort key (-count, value)), ranks 1..n, '<unk>' = 0; emitted id = rank + column offset, offset_0 = 13
  * numeric field i (1..13) -> id i, value (v - min)/(max - min) where min/max were taken over CLIPPED values but v is
    not clipped at emit time; printed "%.6f" with trailing zeros (and a bare '.') stripped; empty -> 0
  * feature_map lists categorical ids +1 relative to the libsvm files (reference quirk, :125 vs :144)
  * tr/va split: random.seed(0); randint(0,9999) % 10 != 0 -> train
  * te.libsvm carries the label of the LAST train line (reference quirk, :167); test columns are shifted by one
There is no hashing anywhere in the reference; ids are dictionary ranks.
Design difference: one pass gathers both the min/max and the 26 count tables (the reference reads train.txt twice
for that), and lines are formatted with precomputed per-field string tables.
"""
from __future__ import annotations

import argparse
import collections
import random
import sys
from typing import Dict, List

N_CONT, N_CAT = 13, 26
CONT_CLIP = [20, 600, 100, 50, 64000, 500, 100, 50, 500, 10, 10, 10, 50]     # get_criteo_feature.py:24


class CriteoStats:
    """min/max of the clipped integer features and the per-column value counts (one pass over train.txt)."""

    def __init__(self):
        self.min = [sys.maxsize] * N_CONT
        self.max = [-sys.maxsize] * N_CONT
        self.counts: List[Dict[str, int]] = [collections.defaultdict(int) for _ in range(N_CAT)]

    def update(self, cols: List[str]) -> None:
        for i in range(N_CONT):
            v = cols[1 + i]
            if v != "":
                x = int(v)
                if x > CONT_CLIP[i]:
                    x = CONT_CLIP[i]
                if x < self.min[i]:
                    self.min[i] = x
                if x > self.max[i]:
                    self.max[i] = x
        for c in range(N_CAT):
            v = cols[14 + c]
            if v != "":
                self.counts[c][v] += 1

    def vocabularies(self, cutoff: int) -> List[Dict[str, int]]:
        dicts = []
        for c in range(N_CAT):
            kept = sorted(((k, n) for k, n in self.counts[c].items() if n >= cutoff), key=lambda kv: (-kv[1], kv[0]))
            if not kept:
                # the reference does `vocabs, _ = list(zip(*[]))` here and dies with ValueError
                raise ValueError("not enough values to unpack (expected 2, got 0)")
            d = {k: r for r, (k, _n) in enumerate(kept, start=1)}
            d["<unk>"] = 0
            dicts.append(d)
        return dicts


def _fmt(v: float) -> str:
    return "{0:.6f}".format(v).rstrip("0").rstrip(".")


def preprocess(input_dir: str, output_dir: str, cutoff: int = 200) -> int:
    """Returns feature_size (= number of lines of feature_map)."""
    stats = CriteoStats()
    with open(input_dir + "train.txt", "r") as f:
        for line in f:
            stats.update(line.rstrip("\n").split("\t"))
    dicts = stats.vocabularies(cutoff)
    offsets = [N_CONT]
    for c in range(N_CAT):
        offsets.append(offsets[c] + len(dicts[c]))
    span = [stats.max[i] - stats.min[i] for i in range(N_CONT)]

    with open(output_dir + "feature_map", "w") as out:
        for i in range(1, N_CONT + 1):
            out.write("I%d %d\n" % (i, i))
        for c in range(N_CAT):
            for key, val in dicts[c].items():
                out.write("C%d|%s %d\n" % (c + 1, key, offsets[c] + val + 1))

    def features(cols: List[str], shift: int) -> str:
        toks = []
        for i in range(N_CONT):
            v = cols[1 + i - shift]
            x = 0.0 if v == "" else (float(v) - stats.min[i]) / span[i]
            toks.append("%d:%s" % (i + 1, _fmt(x)))
        for c in range(N_CAT):
            toks.append("%d:1" % (dicts[c].get(cols[14 + c - shift], 0) + offsets[c]))
        return " ".join(toks)

    random.seed(0)
    label = None
    with open(output_dir + "tr.libsvm", "w") as tr, open(output_dir + "va.libsvm", "w") as va, \
            open(input_dir + "train.txt", "r") as f:
        for line in f:
            cols = line.rstrip("\n").split("\t")
            label = cols[0]
            rec = "%s %s\n" % (label, features(cols, 0))
            (tr if random.randint(0, 9999) % 10 != 0 else va).write(rec)
    with open(output_dir + "te.libsvm", "w") as te, open(input_dir + "test.txt", "r") as f:
        for line in f:
            cols = line.rstrip("\n").split("\t")
            te.write("%s %s\n" % (label, features(cols, 1)))
    return offsets[-1]


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--input_dir", type=str, default="")
    ap.add_argument("--output_dir", type=str, default="")
    ap.add_argument("--cutoff", type=int, default=200)
    a, _ = ap.parse_known_args(argv)
    n = preprocess(a.input_dir, a.output_dir, a.cutoff)
    print("feature_size", n)


if __name__ == "__main__":
    main()
