"""CPU oracle for the variable-length (multi-hot) models -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as
oracle/deepctr_oracle.py: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it).

Restates, op for op in torch-CPU, the graphs built by
  * deep_ctr/Model_pipeline/DIN.py:100-252   (model "din": field-wise SUM pooling, --attention_pooling=False, DIN.py:179-183;
                                              with Config.attention_layers non-empty: the attention units of DIN.py:151-177)
  * DeepMTL/Model_pipeline/DeepCvrMTL.py:107-238 (model "esmm": shared embeddings, CTR and CVR towers, pCTCVR = pCTR * pCVR)
PARITY STATUS: parity unpinned (TensorFlow 1.4 is not available; the reference ships no golden vectors for these scripts).

Inputs follow the scripts' feature dict (DIN.py:59-76 / DeepCvrMTL.py:63-80):
  feat_ids [B, F'] int           common one-hot fields                  -> tf.nn.embedding_lookup           (DIN.py:144)
  u_cat / u_shop / u_brand / u_int : (offsets [B+1], ids [nnz], vals [nnz])  multi-hot with weights
                                                                        -> embedding_lookup_sparse(sum)     (DIN.py:180-183)
  a_cat / a_shop / a_brand [B] int single ids                           -> embedding_lookup                 (DIN.py:145-147)
  a_int : (offsets, ids, None)  multi-hot, no weights                   -> embedding_lookup_sparse(sum)     (DIN.py:148)
and the MLP input is the concat, in this order (DIN.py:199 / DeepCvrMTL.py:165):
  [common (F'*K) | u_cat | u_shop | u_brand | u_int | a_cat | a_shop | a_brand | a_int]   = (F' + 8) slots of K floats.
embedding_lookup_sparse [TF-1.4]: out[b] = sum_j w_j E[id_j] over the row's entries; a row without entries gives zeros
(TF would drop trailing empty rows from the segment_sum output; batches here always end in a non-empty row or rely on zeros).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Sequence

import numpy as np
import torch

from oracle.deepctr_oracle import Optimizer, _dropout, _fc, l2_loss  # noqa: F401  (the TF-1.4 optimizer rules are shared)

MULTI_W = ("u_cat", "u_shop", "u_brand", "u_int")      # multi-hot with weights
SINGLE = ("a_cat", "a_shop", "a_brand")
MULTI_NW = ("a_int",)                                    # multi-hot without weights
LOG_LOSS_EPS = 1e-7                                      # tf.losses.log_loss default epsilon [TF-1.4]


@dataclass
class Config:
    model: str = "din"                   # "din" | "esmm"
    field_size: int = 10                 # F': common one-hot fields (--field_size)
    feature_size: int = 1000
    embedding_size: int = 8
    deep_layers: Sequence[int] = (32, 16)
    dropout: Sequence[float] = (1.0, 1.0)
    l2_reg: float = 1e-4
    learning_rate: float = 5e-4
    optimizer: str = "Adam"
    ctr_task_wgt: float = 0.5            # DeepCvrMTL.py:47
    # DIN attention pooling (DIN.py:45,151-177): widths of the attention MLP; () = field-wise sum pooling.  NOTE the script sizes
    # these layers with layers[i] (the DEEP layer widths, DIN.py:164) for i < len(attention_layers): callers restating the
    # script pass deep_layers[:len(attention_layers)] here
    attention_layers: Sequence[int] = ()
    batch_norm: bool = False             # --batch_norm: batch_norm_layer after every hidden ReLU (DIN.py:203-204, DeepCvrMTL.py:177-178)
    batch_norm_decay: float = 0.9
    batch_norm_bessel: bool = True       # the moving variance is fed var * B/(B-1): TF-1.4's fused batch_norm (rank-2 inputs) [TF-1.4]

    @property
    def n_slots(self) -> int:
        return self.field_size + len(MULTI_W) + len(SINGLE) + len(MULTI_NW)


def towers(cfg: Config) -> List[str]:
    return ["ctr_", "cvr_"] if cfg.model == "esmm" else [""]


def param_shapes(cfg: Config) -> Dict[str, tuple]:
    """Engine names; TF names: embeddings (DIN.py:119), MLP-layer/mlp%d + DIN-out/din_out (DIN.py:201,208);
    cvr_mlp%d / cvr_out / ctr_mlp%d / ctr_out (DeepCvrMTL.py:174,182,195,203)."""
    shp = {"emb": (cfg.feature_size, cfg.embedding_size)}
    for t in towers(cfg):
        d = cfg.n_slots * cfg.embedding_size
        for i, h in enumerate(cfg.deep_layers):
            shp[f"{t}mlp{i}/weights"] = (d, h)
            shp[f"{t}mlp{i}/biases"] = (h,)
            if cfg.batch_norm:           # scopes bn_%d (DIN.py:204) / cvr_bn_%d, ctr_bn_%d (DeepCvrMTL.py:178,199)
                for nm in ("beta", "gamma", "moving_mean", "moving_variance"):
                    shp[f"{t}bn_{i}/{nm}"] = (h,)
            d = h
        out = f"{t}out" if t else "deep_out"
        shp[f"{out}/weights"] = (d, 1)
        shp[f"{out}/biases"] = (1,)
    if cfg.attention_layers:                 # Field-wise-Pooling-layer/att_fc%d, att_out -- shared by the four units (AUTO_REUSE, DIN.py:150)
        d = 3 * cfg.embedding_size
        for i, h in enumerate(cfg.attention_layers):
            shp[f"att_fc{i}/weights"] = (d, h)
            shp[f"att_fc{i}/biases"] = (h,)
            d = h
        shp["att_out/weights"] = (d, 1)
        shp["att_out/biases"] = (1,)
    return shp


def init_params(cfg: Config, seed: int = 0, scale: float = 0.05) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    p = {n: (torch.randn(s, generator=g) * scale).to(torch.float32) for n, s in param_shapes(cfg).items()}
    for n in p:                          # gamma / moving_variance around one, as a trained model would hold them
        if n.endswith(("gamma", "moving_variance")):
            p[n] = 1.0 + p[n].abs()
    return p


def _lookup_sparse(E, offsets, ids, vals):
    """tf.nn.embedding_lookup_sparse(E, sp_ids, sp_weights, combiner="sum") (DIN.py:180): gather, weight, segment_sum."""
    offsets = np.asarray(offsets)
    B = len(offsets) - 1
    rows = E[torch.as_tensor(np.asarray(ids), dtype=torch.long)]
    if vals is not None:
        rows = rows * torch.as_tensor(np.asarray(vals), dtype=E.dtype)[:, None]
    seg = torch.as_tensor(np.repeat(np.arange(B), np.diff(offsets)), dtype=torch.long)
    return torch.zeros((B, E.shape[1]), dtype=E.dtype).index_add(0, seg, rows)


ATT_PAIRS = (("u_cat", "a_cat"), ("u_shop", "a_shop"), ("u_brand", "a_brand"), ("u_int", "a_int"))      # DIN.py:174-177


def _attention_unit(cfg: Config, p, E, offsets, ids, vals, a_emb, train, masks, key):
    """attention_unit (DIN.py:152-172) over the row's real entries: the padded positions of the dense [B, P] form have
    dense_emb = E[0] * 0 = 0, so they add nothing to the output nor to any gradient.
      ub = w * E[id];  x = [ub, ub - ax, ax];  att = sigmoid(fc(relu-MLP(x)));  out[b] = sum_j ub_j att_j [id_j > 0]"""
    offsets = np.asarray(offsets)
    B = len(offsets) - 1
    idt = torch.as_tensor(np.asarray(ids), dtype=torch.long)
    seg = torch.as_tensor(np.repeat(np.arange(B), np.diff(offsets)), dtype=torch.long)
    ub = E[idt] * torch.as_tensor(np.asarray(vals), dtype=E.dtype)[:, None]
    ax = a_emb[seg]
    x = torch.cat([ub, ub - ax, ax], dim=1)
    for i in range(len(cfg.attention_layers)):
        x = _fc(x, p[f"att_fc{i}/weights"], p[f"att_fc{i}/biases"])
        x = _dropout(x, cfg.dropout[i], train, masks, f"{key}/att_fc{i}")
    att = torch.sigmoid(_fc(x, p["att_out/weights"], p["att_out/biases"], relu=False))          # [n, 1]
    mask = (idt > 0).to(E.dtype)[:, None]                                                        # DIN.py:157: id 0 = padding
    return torch.zeros((B, E.shape[1]), dtype=E.dtype).index_add(0, seg, ub * att * mask)


def embed(cfg: Config, p, batch, train: bool = False, masks=None) -> torch.Tensor:
    """x_concat [B, (F'+8)*K] (DIN.py:143-148,150-183,199; DeepCvrMTL.py:153-165)."""
    E = p["emb"]
    feat_ids = torch.as_tensor(np.asarray(batch["feat_ids"]), dtype=torch.long)
    B = feat_ids.shape[0]
    ad = {n: E[torch.as_tensor(np.asarray(batch[n]), dtype=torch.long)] for n in SINGLE}
    for n in MULTI_NW:
        off, ids, _ = batch[n]
        ad[n] = _lookup_sparse(E, off, ids, None)
    parts = [E[feat_ids].reshape(B, -1)]
    for u, a in ATT_PAIRS:
        if cfg.attention_layers:
            parts.append(_attention_unit(cfg, p, E, *batch[u], ad[a], train, masks, u))
        else:
            parts.append(_lookup_sparse(E, *batch[u]))
    parts += [ad[n] for n in SINGLE + MULTI_NW]
    return torch.cat(parts, dim=1)


def _bn(cfg: Config, p, x, scope, train, new_stats):
    """contrib.layers.batch_norm(decay, center, scale, eps=1e-3), moving averages updated in place in TRAIN [TF-1.4]
    (batch_norm_layer, DIN.py:254-258)."""
    g, bt, mm, mv = (p[f"{scope}/{n}"] for n in ("gamma", "beta", "moving_mean", "moving_variance"))
    if train:
        mean, var = x.mean(0), x.var(0, unbiased=False)
        d = cfg.batch_norm_decay
        new_stats[f"{scope}/moving_mean"] = d * mm + (1 - d) * mean.detach()
        n = x.shape[0]
        adj = n / max(n - 1, 1) if cfg.batch_norm_bessel else 1.0        # fused_batch_norm_op.cc `rest_size_adjust` [TF-1.4]
        new_stats[f"{scope}/moving_variance"] = d * mv + (1 - d) * var.detach() * adj
    else:
        mean, var = mm, mv
    return (x - mean) / torch.sqrt(var + 1e-3) * g + bt


def _tower(cfg: Config, p, x, prefix, train, masks, new_stats=None):
    new_stats = {} if new_stats is None else new_stats
    for i in range(len(cfg.deep_layers)):
        x = _fc(x, p[f"{prefix}mlp{i}/weights"], p[f"{prefix}mlp{i}/biases"])
        if cfg.batch_norm:
            x = _bn(cfg, p, x, f"{prefix}bn_{i}", train, new_stats)
        x = _dropout(x, cfg.dropout[i], train, masks, f"{prefix}mlp{i}")
    out = f"{prefix}out" if prefix else "deep_out"
    return _fc(x, p[f"{out}/weights"], p[f"{out}/biases"], relu=False).reshape(-1)


def forward(cfg: Config, p, batch, train: bool = False, masks=None):
    x = embed(cfg, p, batch, train, masks)
    ns: Dict[str, torch.Tensor] = {}
    if cfg.model == "din":
        y = _tower(cfg, p, x, "", train, masks, ns)                  # DIN.py:200-210
        return {"x": x, "y": y, "prob": torch.sigmoid(y), "_new_stats": ns}
    y_cvr = _tower(cfg, p, x, "cvr_", train, masks, ns)              # DeepCvrMTL.py:167-184
    y_ctr = _tower(cfg, p, x, "ctr_", train, masks, ns)              # DeepCvrMTL.py:186-205
    pctr, pcvr = torch.sigmoid(y_ctr), torch.sigmoid(y_cvr)          # DeepCvrMTL.py:207-210
    return {"x": x, "y_ctr": y_ctr, "y_cvr": y_cvr, "pctr": pctr, "pcvr": pcvr, "pctcvr": pctr * pcvr, "_new_stats": ns}


def _xent(y, z):
    return torch.clamp(y, min=0) - y * z + torch.log1p(torch.exp(-y.abs()))     # sigmoid_cross_entropy_with_logits [TF-1.4]


def loss_fn(cfg: Config, p, out, batch):
    yl = torch.as_tensor(np.asarray(batch["y"]), dtype=torch.float32)
    if cfg.model == "din":                                           # DIN.py:222
        return _xent(out["y"], yl).mean() + cfg.l2_reg * l2_loss(p["emb"])
    zl = torch.as_tensor(np.asarray(batch["z"]), dtype=torch.float32)
    ctr_loss = _xent(out["y_ctr"], yl).mean()                        # DeepCvrMTL.py:222
    pr = out["pctcvr"]                                               # tf.losses.log_loss (eps 1e-7, mean over the batch) [TF-1.4]
    ll = -zl * torch.log(pr + LOG_LOSS_EPS) - (1 - zl) * torch.log(1 - pr + LOG_LOSS_EPS)
    cvr_loss = ll.mean()                                             # DeepCvrMTL.py:224
    w = cfg.ctr_task_wgt
    return w * ctr_loss + (1 - w) * cvr_loss + cfg.l2_reg * l2_loss(p["emb"])      # DeepCvrMTL.py:225


def grads(cfg: Config, p, batch, train=True, masks=None):
    names = [n for n in p if not n.endswith(("moving_mean", "moving_variance"))]
    leaves = {n: p[n].detach().clone().requires_grad_(True) for n in names}
    q = dict(p)
    q.update(leaves)
    out = forward(cfg, q, batch, train=train, masks=masks)
    loss = loss_fn(cfg, q, out, batch)
    gs = torch.autograd.grad(loss, [leaves[n] for n in names])
    return loss.detach(), dict(zip(names, gs)), out


def train_step(cfg: Config, p, opt: Optimizer, batch, masks=None):
    loss, g, out = grads(cfg, p, batch, train=True, masks=masks)
    for k, v in out["_new_stats"].items():
        p[k] = v
    opt.step(p, g)
    return float(loss), out


# ---- synthetic batches + the slot-ordered CSR the engine consumes -------------------------------------------------------------
def synth_batch(cfg: Config, B: int, seed: int = 0, max_len: int = 6, allow_empty: bool = True):
    rng = np.random.default_rng(seed)
    V = cfg.feature_size
    batch = {"feat_ids": rng.integers(1, V, size=(B, cfg.field_size)).astype(np.int64)}

    def multi(weights: bool):
        lens = rng.integers(0 if allow_empty else 1, max_len + 1, size=B)
        lens[-1] = max(lens[-1], 1)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        # a zipf-ish head so that ids repeat inside a batch (segment sums with several contributors)
        ids = np.minimum(rng.zipf(1.3, size=int(off[-1])), V - 1).astype(np.int64)
        vals = rng.uniform(0.5, 3.0, size=int(off[-1])).astype(np.float32) if weights else None
        return off, ids, vals

    for n in MULTI_W:
        batch[n] = multi(True)
    for n in SINGLE:
        batch[n] = rng.integers(1, V, size=B).astype(np.int64)
    for n in MULTI_NW:
        batch[n] = multi(False)
    batch["y"] = (rng.random(B) < 0.3).astype(np.float32)
    batch["z"] = (batch["y"] * (rng.random(B) < 0.4)).astype(np.float32)        # a conversion implies a click
    return batch


def slot_csr(cfg: Config, batch):
    """(offsets int32 [B*S+1], ids int32 [nnz], weights f32 [nnz]) with segment b*S+s = slot s of example b, slots in the
    concat order of DIN.py:199 (single-id slots = one entry of weight 1)."""
    B = batch["feat_ids"].shape[0]
    S = cfg.n_slots
    ids: List[int] = []
    wts: List[float] = []
    offsets = [0]
    for b in range(B):
        for f in range(cfg.field_size):
            ids.append(int(batch["feat_ids"][b, f])); wts.append(1.0); offsets.append(len(ids))
        for n in MULTI_W:
            off, i, v = batch[n]
            ids.extend(int(t) for t in i[off[b]:off[b + 1]]); wts.extend(float(t) for t in v[off[b]:off[b + 1]]); offsets.append(len(ids))
        for n in SINGLE:
            ids.append(int(batch[n][b])); wts.append(1.0); offsets.append(len(ids))
        for n in MULTI_NW:
            off, i, _ = batch[n]
            ids.extend(int(t) for t in i[off[b]:off[b + 1]]); wts.extend([1.0] * int(off[b + 1] - off[b])); offsets.append(len(ids))
    assert len(offsets) == B * S + 1
    return np.asarray(offsets, np.int32), np.asarray(ids, np.int32), np.asarray(wts, np.float32)
