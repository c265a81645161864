"""CPU oracle for the canned-estimator path of wide_n_deep.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/`` may import this module (same rule as oracle/deepctr_oracle.py).

What it restates: the graphs that ``tf.estimator.LinearClassifier`` / ``DNNClassifier`` /
``DNNLinearCombinedClassifier`` build for the feature columns of wide_n_deep.py:92-105 and the estimators of
wide_n_deep.py:113-151.  All arithmetic lives inside TensorFlow 1.4 (not vendored, not installable here); the reference
pins nothing for this path, so every rule below is tagged [TF-1.4] and is the restater's reading of the TF 1.4 sources:
PARITY STATUS: **parity unpinned** (SURVEY.md section 8a row a15, section 8c).

  columns   13 numeric_column I1..I13 (wide_n_deep.py:94) and 26 categorical_column_with_identity(num_buckets=10000,
            default_value=0) C14..C39 (:97-98): ids outside [0, 10000) become 0 [TF-1.4 identity column].
  wide      linear_model: one weight per numeric column, one [10000,1] table per categorical column (sparse combiner
            "sum"), one bias; all zero-initialised [TF-1.4 feature_column.linear_model].
  deep      input_layer concatenates the columns in NAME-SORTED order [TF-1.4 feature_column.input_layer]:
            C14_embedding..C39_embedding (26*K) then I1, I10, I11, I12, I13, I2..I9; embedding rows ~ truncated
            normal(0, 1/sqrt(K)), combiner "mean" (one id per example: the row itself); hidden layers
            relu(x W + b) with glorot-uniform W, zero b; logits layer [H,1] + bias [TF-1.4 dnn._dnn_logit_fn].
            This module keeps the 13 numeric inputs in natural order I1..I13 -- the name-sorted permutation only
            reorders rows of the first kernel and is applied where TF variable names are mapped (tf_shim).
  logits    dnn_logits + linear_logits [TF-1.4 dnn_linear_combined._dnn_linear_combined_model_fn].
  loss      SUM over the batch of sigmoid cross-entropy [TF-1.4 head._binary_logistic_head_with_sigmoid_cross_entropy_loss,
            losses.Reduction.SUM]; no regularisation.
  optimizers (defaults, wide_n_deep.py passes none) [TF-1.4]:
            LinearClassifier             Ftrl(lr = min(0.2, 1/sqrt(#columns)))          (linear.py _LEARNING_RATE = 0.2)
            DNNClassifier                Adagrad(lr = 0.05)                             (dnn.py _LEARNING_RATE = 0.05)
            DNNLinearCombinedClassifier  dnn: Adagrad(0.001); linear: Ftrl(min(0.005, 1/sqrt(#linear columns)))
                                         (dnn_linear_combined.py _DNN_LEARNING_RATE = 0.001, _LINEAR_LEARNING_RATE = 0.005:
                                         "a historical artifact of the initial implementation")
            Adagrad and Ftrl start their accumulators at 0.1; Ftrl lr_power -0.5, no l1/l2.  Table gradients are
            IndexedSlices: duplicates are summed and ONLY the touched rows are updated (sparse apply).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Sequence

import numpy as np
import torch

NUM_BUCKETS = 10000          # wide_n_deep.py:97
N_NUMERIC = 13               # wide_n_deep.py:55
N_CATEGORICAL = 26           # wide_n_deep.py:56


@dataclass
class CannedConfig:
    model_type: str = "wide_n_deep"              # wide | deep | wide_n_deep   (wide_n_deep.py:47)
    embedding_size: int = 32                     # wide_n_deep.py:31
    deep_layers: Sequence[int] = (256, 128, 64)  # wide_n_deep.py:34
    num_buckets: int = NUM_BUCKETS
    n_numeric: int = N_NUMERIC
    n_categorical: int = N_CATEGORICAL

    @property
    def wide(self) -> bool:
        return self.model_type in ("wide", "wide_n_deep")

    @property
    def deep(self) -> bool:
        return self.model_type in ("deep", "wide_n_deep")

    @property
    def rows(self) -> int:
        return self.n_categorical * self.num_buckets

    @property
    def dnn_learning_rate(self) -> float:
        return 0.05 if self.model_type == "deep" else 0.001

    @property
    def linear_learning_rate(self) -> float:
        cap = 0.2 if self.model_type == "wide" else 0.005
        return min(cap, 1.0 / math.sqrt(self.n_numeric + self.n_categorical))


def table_rows(cfg: CannedConfig, cat_ids: np.ndarray) -> np.ndarray:
    """[B,26] raw categorical values -> rows of the stacked per-column tables: column c owns rows [c*10000, (c+1)*10000);
    out-of-range ids map to default_value 0 (wide_n_deep.py:97-98)."""
    ids = np.asarray(cat_ids, dtype=np.int64)
    ids = np.where((ids >= 0) & (ids < cfg.num_buckets), ids, 0)
    return (ids + np.arange(cfg.n_categorical, dtype=np.int64)[None, :] * cfg.num_buckets).astype(np.int32)


def param_shapes(cfg: CannedConfig) -> Dict[str, tuple]:
    s: Dict[str, tuple] = {}
    if cfg.wide:
        s["bias"] = (1,)
        s["linear"] = (cfg.rows,)
        s["linear_dense"] = (cfg.n_numeric,)
    if cfg.deep:
        s["emb"] = (cfg.rows, cfg.embedding_size)
        d = cfg.n_categorical * cfg.embedding_size + cfg.n_numeric
        for i, h in enumerate(cfg.deep_layers):
            s["mlp%d/weights" % i] = (d, h)
            s["mlp%d/biases" % i] = (h,)
            d = h
        s["deep_out/weights"] = (d, 1)
        s["deep_out/biases"] = (1,)
    return s


def init_params(cfg: CannedConfig, seed: int = 0, scale: float = 0.05) -> Dict[str, torch.Tensor]:
    """Random (not TF's) initial values: parity tests inject the same tensors on both sides."""
    g = torch.Generator().manual_seed(seed)
    return {k: torch.randn(*shp, generator=g) * scale for k, shp in param_shapes(cfg).items()}


def forward(cfg: CannedConfig, p, rows, numeric) -> torch.Tensor:
    """rows int64 [B,26] (table_rows), numeric f32 [B,13] -> logits [B]."""
    B = rows.shape[0]
    y = torch.zeros(B, dtype=numeric.dtype)
    if cfg.wide:
        y = y + p["bias"] + p["linear"][rows].sum(1) + numeric @ p["linear_dense"]
    if cfg.deep:
        x = torch.cat([p["emb"][rows].reshape(B, -1), numeric], dim=1)
        for i in range(len(cfg.deep_layers)):
            x = torch.relu(x @ p["mlp%d/weights" % i] + p["mlp%d/biases" % i])
        y = y + (x @ p["deep_out/weights"]).reshape(-1) + p["deep_out/biases"]
    return y


def loss_fn(y, labels) -> torch.Tensor:
    """sum_b [max(y,0) - y z + log(1 + exp(-|y|))]   [TF-1.4 sigmoid_cross_entropy_with_logits, Reduction.SUM]"""
    return (torch.clamp(y, min=0) - y * labels + torch.log1p(torch.exp(-torch.abs(y)))).sum()


WIDE_PARAMS = ("bias", "linear", "linear_dense")


class CannedOptimizer:
    """Ftrl for the linear side, Adagrad for the DNN side, sparse apply on the tables [TF-1.4]."""

    def __init__(self, cfg: CannedConfig, p):
        self.cfg = cfg
        self.accum = {n: torch.full_like(v, 0.1) for n, v in p.items()}
        self.ftrl_linear = {n: torch.zeros_like(v) for n, v in p.items() if n in WIDE_PARAMS}

    def step(self, p, g, rows) -> None:
        touched = torch.unique(rows.reshape(-1))
        for n, gn in g.items():
            table = n in ("linear", "emb")
            idx = touched if table else slice(None)
            th, gi, acc = p[n][idx], gn[idx], self.accum[n][idx]
            if n in WIDE_PARAMS:
                lr = self.cfg.linear_learning_rate
                new_acc = acc + gi * gi
                sigma = (torch.sqrt(new_acc) - torch.sqrt(acc)) / lr
                lin = self.ftrl_linear[n][idx] + gi - sigma * th
                new = -lin / (torch.sqrt(new_acc) / lr)
                self.ftrl_linear[n][idx] = lin
            else:
                lr = self.cfg.dnn_learning_rate
                new_acc = acc + gi * gi
                new = th - lr * gi / torch.sqrt(new_acc)
            self.accum[n][idx] = new_acc
            p[n] = p[n].clone()
            p[n][idx] = new


def train_step(cfg: CannedConfig, p, opt: CannedOptimizer, rows, numeric, labels) -> float:
    names = list(p)
    leaves = {n: p[n].detach().clone().requires_grad_(True) for n in names}
    loss = loss_fn(forward(cfg, leaves, rows, numeric), labels)
    gs = torch.autograd.grad(loss, [leaves[n] for n in names])
    opt.step(p, dict(zip(names, gs)), rows)
    return float(loss.detach())


def synth_csv_batch(B: int, seed: int = 0):
    """labels f32 [B], numeric f32 [B,13], categorical int64 [B,26] (a few values out of range, to exercise default_value)."""
    r = np.random.default_rng(seed)
    labels = (r.random(B) < 0.25).astype(np.float32)
    numeric = np.round(r.random((B, N_NUMERIC)), 6).astype(np.float32)
    cat = r.zipf(1.2, size=(B, N_CATEGORICAL)).astype(np.int64) % 12000 - 3      # some < 0 and some >= 10000
    return labels, numeric, cat
