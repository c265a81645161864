"""TF checkpoint bundles without TensorFlow (tf_repos_amd/tf_bundle.py).  UNPINNED against TF itself (no
TF-written checkpoint exists here): round trips plus the structural invariants of the table format."""
import os
import struct

import numpy as np
import pytest

from tf_repos_amd import errors, tf_bundle as T


def _tensors(rng, n_extra=0):
    t = {"fm_bias": rng.normal(size=(1,)).astype(np.float32), "fm_w": rng.normal(size=(1000,)).astype(np.float32),
         "fm_v": rng.normal(size=(1000, 8)).astype(np.float32), "Deep-part/mlp0/weights": rng.normal(size=(312, 16)).astype(np.float32),
         "Deep-part/mlp0/biases": np.zeros(16, np.float32), "global_step": np.int64(1234),
         "fm_v/Adam": rng.normal(size=(1000, 8)).astype(np.float32), "fm_v/Adam_1": rng.random((1000, 8)).astype(np.float32)}
    for i in range(n_extra):
        t["Deep-part/extra_%03d/weights" % i] = rng.normal(size=(3, 5)).astype(np.float32)
    return t


@pytest.mark.parametrize("n_extra,block_size", [(0, 4096), (300, 256)])        # the second: many data blocks, restart arrays, shared prefixes
def test_round_trip_and_structure(tmp_path, n_extra, block_size):
    t = _tensors(np.random.default_rng(0), n_extra)
    prefix = str(tmp_path / "model.ckpt-1234")
    T.write_bundle(prefix, t, block_size=block_size)
    idx = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", idx[-8:])[0] == 0xdb4775248b80fb57                 # table magic
    back = T.read_bundle(prefix)
    assert set(back) == set(t)
    for k, v in t.items():
        assert back[k].dtype == np.asarray(v).dtype and np.array_equal(back[k], np.asarray(v)), k
    assert os.path.getsize(prefix + ".data-00000-of-00001") == sum(np.asarray(v).nbytes for v in t.values())
    assert T.latest_tf_checkpoint(str(tmp_path)) == prefix


def test_corruption_is_a_data_loss_error(tmp_path):
    t = _tensors(np.random.default_rng(1))
    prefix = str(tmp_path / "model.ckpt-1")
    T.write_bundle(prefix, t)
    data = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    data[100] ^= 0xFF
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    with pytest.raises(errors.DataLossError):
        T.read_bundle(prefix)
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[10] ^= 0xFF
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(errors.DataLossError):
        T.read_index(prefix + ".index")


def test_slot_names_follow_tf(tmp_path):
    t = _tensors(np.random.default_rng(2))
    st = T.bundle_to_state(t, "Adam")
    assert "fm_v/slot0" in st and "fm_v/slot1" in st and "fm_v/Adam" not in st and "fm_w" in st
    back = T.state_to_bundle(st, "Adam")
    # AdamOptimizer's non-slot variables travel too (a TF training graph's Saver asks for them): beta^(global_step + 1): TF starts them at beta and multiplies after each apply
    assert set(back) == set(t) | {"beta1_power", "beta2_power"} and np.array_equal(back["fm_v/Adam_1"], t["fm_v/Adam_1"])
    gs = int(np.asarray(t["global_step"]))
    assert abs(float(back["beta1_power"]) - 0.9 ** (gs + 1)) < 1e-6 and abs(float(back["beta2_power"]) - 0.999 ** (gs + 1)) < 1e-6
    assert "beta1_power" not in T.bundle_to_state(back, "Adam")
    # a checkpoint of an untrained model: TF's initial values (beta, not 1 -- lr_t = lr sqrt(1 - beta2_power) / (1 - beta1_power) must be finite)
    fresh = T.state_to_bundle({"w": np.ones(3, np.float32), "global_step": np.int64(0)}, "Adam")
    assert float(fresh["beta1_power"]) == np.float32(0.9) and float(fresh["beta2_power"]) == np.float32(0.999)
    custom = T.state_to_bundle({"w": np.ones(3, np.float32), "global_step": np.int64(2)}, "Adam", beta1=0.8, beta2=0.99)
    assert abs(float(custom["beta1_power"]) - 0.8 ** 3) < 1e-7 and abs(float(custom["beta2_power"]) - 0.99 ** 3) < 1e-7
    ftrl = T.state_to_bundle({"w": np.ones(3, np.float32), "w/slot0": np.ones(3, np.float32), "w/slot1": np.zeros(3, np.float32)}, "ftrl")
    assert set(ftrl) == {"w", "w/Ftrl", "w/Ftrl_1"}
    assert set(T.state_to_bundle({"w": np.ones(3, np.float32), "w/slot0": np.ones(3, np.float32), "w/slot1": np.zeros(3, np.float32)}, "Adagrad")) == {"w", "w/Adagrad"}


def test_snappy_blocks_are_read():
    # literal + copy elements (the index of a bundle written with compression on)
    raw = b"abcdabcdabcd" + b"xyz"
    comp = bytes([len(raw)]) + bytes([(4 - 1) << 2]) + b"abcd" + bytes([((8 - 4) << 2) | 1, 4]) + bytes([(3 - 1) << 2]) + b"xyz"
    assert T._snappy_decompress(comp) == raw
