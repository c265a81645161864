"""Integer bucketing must be BIT-EXACT (BASELINE.json north_star).  Golden fixtures = the reference's own
get_criteo_feature.py run on a synthetic Criteo TSV (tests/golden/make_bucketing_golden.py)."""
import filecmp
import os

import pytest

from oracle import bucketing_oracle as BO
from tf_repos_amd import criteo_features as CF

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "criteo_small")


def test_product_pipeline_is_byte_identical_to_reference_output(tmp_path):
    out = str(tmp_path) + "/"
    n = CF.preprocess(GOLD + "/", out, cutoff=3)
    for f in ["tr.libsvm", "va.libsvm", "te.libsvm", "feature_map"]:
        assert filecmp.cmp(os.path.join(GOLD, f), out + f, shallow=False), f
    assert n == sum(1 for _ in open(os.path.join(GOLD, "feature_map")))       # feature_size = #lines of feature_map


def test_reference_quirks_are_preserved():
    lines = open(os.path.join(GOLD, "tr.libsvm")).read().splitlines()
    toks = lines[0].split(" ")
    assert [t.split(":")[0] for t in toks[1:14]] == [str(i) for i in range(1, 14)]    # numeric field i -> id i
    # id 13 is shared by I13 and C1's <unk> (rank 0 + offset 13)
    te = open(os.path.join(GOLD, "te.libsvm")).read()
    assert " 13:" in te
    # te.libsvm label = label of the last train line (stale variable in the reference)
    last_label = open(os.path.join(GOLD, "train.txt")).read().splitlines()[-1].split("\t")[0]
    assert all(l.split(" ")[0] == last_label for l in te.splitlines())
    # feature_map ids are +1 relative to the libsvm ids
    fmap = dict(l.rsplit(" ", 1) for l in open(os.path.join(GOLD, "feature_map")).read().splitlines())
    assert fmap["C1|<unk>"] == "14"


def test_oracle_restatement_matches_golden():
    train = open(os.path.join(GOLD, "train.txt")).read().splitlines()
    mins, maxs, dicts, offsets = BO.build(train, cutoff=3)
    import random
    random.seed(0)
    tr, va = [], []
    for line in train:
        feats = line.split("\t")
        ids, vals = BO.encode(feats, mins, maxs, dicts, offsets)
        rec = feats[0] + " " + " ".join("%d:%s" % iv for iv in zip(ids, vals))
        (tr if random.randint(0, 9999) % 10 != 0 else va).append(rec)
    assert tr == open(os.path.join(GOLD, "tr.libsvm")).read().splitlines()
    assert va == open(os.path.join(GOLD, "va.libsvm")).read().splitlines()


def test_empty_vocabulary_dies_like_the_reference(tmp_path):
    with pytest.raises(ValueError):
        CF.preprocess(GOLD + "/", str(tmp_path) + "/", cutoff=10 ** 6)
