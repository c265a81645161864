"""CPU oracle for the deep_ctr hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module.  The product path (``tf_repos_amd``) never imports it and
fails loudly when the HIP library is missing.

What it is: a torch-CPU (fp32, with an fp64 shadow via ``dtype=``) restatement, op for
op, of the TF-1.4 graph that the reference's ``model_fn``s build.  Each function cites
the reference file:line it follows (paths relative to /root/reference).

PARITY STATUS: pinned to the reference's own SOURCE since round 2, not to TensorFlow.  TensorFlow 1.4 is not vendored, not
installed and not installable here, and the reference ships no tests / golden vectors / checkpoints for this path (SURVEY.md
section 8c) -- but every ``tf.*`` call the reference's ``model_fn``s make is recorded as a symbolic graph and evaluated in numpy
fp64 by ``oracle/graph_eval.py``; ``tests/golden/make_model_golden.py`` writes the results (logits, loss, every gradient, every
variable after two optimizer steps) as fixtures under ``tests/golden/models/`` and ``tests/test_model_golden.py`` holds THIS module to
them (1e-9 in fp64, 2e-5 in fp32) for all eight models, the four optimizers, batch-norm and dropout.  What stays ASSUMED (and is
tagged [TF-1.4] below): the per-op semantics of TF 1.4 itself -- restated from its documentation, never run.
The *integer bucketing* side is pinned bit for bit: ``oracle/bucketing_oracle.py`` against the reference's own
``get_criteo_feature.py`` executed in this container (fixtures in tests/golden/).
"""
from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

MODELS = ("deepfm", "fnn", "ipnn", "opnn", "nfm", "afm", "dcn", "mvm")


@dataclass
class Config:
    """Hyper-parameters, named after the reference flags (DeepFM.py:34-60)."""
    model: str = "deepfm"
    field_size: int = 39
    feature_size: int = 117581
    embedding_size: int = 8
    deep_layers: Sequence[int] = (400, 400, 400)
    dropout: Sequence[float] = (1.0, 1.0, 1.0)      # TF keep_prob (DeepFM.py:162)
    attention_layers: Sequence[int] = (256,)        # AFM.py:52
    cross_layers: int = 3                           # DCN.py:52
    l2_reg: float = 1e-4
    learning_rate: float = 5e-4
    optimizer: str = "Adam"                         # Adam | Adagrad | Momentum | ftrl
    batch_norm: bool = False
    batch_norm_decay: float = 0.9
    batch_norm_bessel: bool = True       # the moving variance is fed var * B/(B-1): TF-1.4's fused batch_norm (rank-2 inputs) [TF-1.4]

    @property
    def num_pairs(self) -> int:
        return self.field_size * (self.field_size - 1) // 2   # PNN.py:113 (py2 int division)

    def has_linear(self) -> bool:
        return self.model not in ("dcn", "mvm")     # DCN / DeepMVM have no linear table (DCN.py:120-125, DeepMVM.py:114-118)

    def mlp_input_width(self) -> int:
        F, K, P = self.field_size, self.embedding_size, self.num_pairs
        return {"deepfm": F * K, "fnn": F * K, "ipnn": F * K + P, "opnn": F * K + P * K * K,
                "nfm": K, "dcn": F * K, "afm": K, "mvm": F * K}[self.model]


# --------------------------------------------------------------------------------------
# parameter shapes / names.  Names are the engine's canonical names; the TF variable name
# each one corresponds to is given in tf_repos_amd.checkpoint.TF_NAMES.
# --------------------------------------------------------------------------------------
def param_shapes(cfg: Config) -> Dict[str, Tuple[int, ...]]:
    F, K, V = cfg.field_size, cfg.embedding_size, cfg.feature_size
    shapes: Dict[str, Tuple[int, ...]] = {}
    if cfg.model == "mvm":
        shapes["mvm_b"] = (F, K)                               # DeepMVM.py:118
        shapes["emb"] = (V, K)                                 # DeepMVM.py:117 (mvm_w)
    elif cfg.model == "dcn":
        shapes["cross_b"] = (cfg.cross_layers, F * K)          # DCN.py:120
        shapes["cross_w"] = (cfg.cross_layers, F * K)          # DCN.py:122
        shapes["emb"] = (V, K)                                 # DCN.py:124
    else:
        shapes["bias"] = (1,)                                  # DeepFM.py:114 / PNN.py:116
        shapes["linear"] = (V,)                                # DeepFM.py:115
        shapes["emb"] = (V, K)                                 # DeepFM.py:116
    if cfg.model == "afm":
        d = K
        for i, a in enumerate(cfg.attention_layers):           # AFM.py:143-145
            shapes[f"att_mlp{i}/weights"] = (d, a)
            shapes[f"att_mlp{i}/biases"] = (a,)
            d = a
        shapes["attention_out/weights"] = (d, 1)               # AFM.py:147
        shapes["attention_out/biases"] = (1,)
        shapes["deep_out/weights"] = (K, 1)                    # AFM.py:160
        shapes["deep_out/biases"] = (1,)
        return shapes
    d = cfg.mlp_input_width()
    for i, h in enumerate(cfg.deep_layers):                    # DeepFM.py:152-158
        shapes[f"mlp{i}/weights"] = (d, h)
        shapes[f"mlp{i}/biases"] = (h,)
        if cfg.batch_norm:                                     # DeepFM.py:159-160,231-235
            shapes[f"bn_{i}/beta"] = (h,)
            shapes[f"bn_{i}/gamma"] = (h,)
            shapes[f"bn_{i}/moving_mean"] = (h,)
            shapes[f"bn_{i}/moving_variance"] = (h,)
        d = h
    if cfg.model == "mvm":
        shapes["deep_out/weights"] = (K + d, 1)                # DeepMVM.py:185-188
        shapes["deep_out/biases"] = (1,)
    elif cfg.model == "dcn":
        shapes["out_layer/weights"] = (F * K + d, 1)           # DCN.py:179-182
        shapes["out_layer/biases"] = (1,)
    else:
        shapes["deep_out/weights"] = (d, 1)                    # DeepFM.py:165-166
        shapes["deep_out/biases"] = (1,)
    return shapes


NON_TRAINABLE_SUFFIXES = ("moving_mean", "moving_variance")


def init_params(cfg: Config, seed: int = 0, scale: float = 0.01, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """N(0, scale) weights (parity tests inject weights; TF's RNG streams are not
    reproducible outside TF, DeepFM.py:114-116)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shp in param_shapes(cfg).items():
        if name.endswith("moving_variance") or name.endswith("gamma"):
            a = np.ones(shp, dtype=np.float32)
        elif name.endswith("moving_mean") or name.endswith("beta"):
            a = np.zeros(shp, dtype=np.float32)
        else:
            a = rng.normal(0.0, scale, size=shp).astype(np.float32)
        out[name] = torch.from_numpy(a).to(dtype)
    return out


# --------------------------------------------------------------------------------------
# forward
# --------------------------------------------------------------------------------------
def pair_index(F: int) -> Tuple[List[int], List[int]]:
    """Lexicographic (i<j) pair order, PNN.py:144-147 / AFM.py:134-136."""
    row, col = [], []
    for i in range(F - 1):
        for j in range(i + 1, F):
            row.append(i)
            col.append(j)
    return row, col


def _fc(x, w, b, relu=True):
    """contrib.layers.fully_connected: y = act(x W + b), act=relu by default [TF-1.4]."""
    y = x @ w + b
    return torch.relu(y) if relu else y


def _bn(x, p, i, cfg, train):
    """contrib.layers.batch_norm(decay, center, scale, eps=1e-3) [TF-1.4]; DeepFM.py:231-235.
    Returns (y, new_moving_mean, new_moving_var)."""
    g, bt = p[f"bn_{i}/gamma"], p[f"bn_{i}/beta"]
    mm, mv = p[f"bn_{i}/moving_mean"], p[f"bn_{i}/moving_variance"]
    if train:
        mean = x.mean(0)
        var = x.var(0, unbiased=False)
        d = cfg.batch_norm_decay
        new_mm = d * mm + (1 - d) * mean.detach()
        n = x.shape[0]
        adj = n / max(n - 1, 1) if cfg.batch_norm_bessel else 1.0       # fused_batch_norm_op.cc `rest_size_adjust` [TF-1.4]
        new_mv = d * mv + (1 - d) * var.detach() * adj
    else:
        mean, var, new_mm, new_mv = mm, mv, mm, mv
    y = (x - mean) / torch.sqrt(var + 1e-3) * g + bt
    return y, new_mm, new_mv


def _dropout(x, keep, train, masks, key):
    """nn.dropout(x, keep) = x * floor(keep + U[0,1)) / keep [TF-1.4]; only in TRAIN
    (DeepFM.py:161-162).  A mask (0/1) may be injected for parity; keep==1 is identity."""
    if not train or keep >= 1.0:
        return x
    if masks is not None and key in masks:
        m = masks[key].to(x.dtype)
    else:
        m = torch.floor(keep + torch.rand_like(x))
    return x * m / keep


def forward(cfg: Config, p: Dict[str, torch.Tensor], ids, vals, train: bool = False,
            masks: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
    """Returns dict with 'y' (logit), 'prob', and named intermediates.  ids [B,F] int64,
    vals [B,F].  Follows DeepFM.py:119-176, PNN.py:121-194, NFM.py:111-156,
    AFM.py:116-168, DCN.py:127-184."""
    F, K = cfg.field_size, cfg.embedding_size
    ids = torch.as_tensor(ids).reshape(-1, F).long()
    dt = p["emb"].dtype
    vals = torch.as_tensor(vals).reshape(-1, F).to(dt)
    B = ids.shape[0]
    if int(ids.min()) < 0 or int(ids.max()) >= cfg.feature_size:
        raise IndexError("feat_ids out of range [0, feature_size) -- TF CPU gather raises "
                         "InvalidArgumentError [TF-1.4]")
    out: Dict[str, torch.Tensor] = {}
    new_stats: Dict[str, torch.Tensor] = {}
    if cfg.has_linear():
        feat_wgts = p["linear"][ids]                            # DeepFM.py:126
        y_w = (feat_wgts * vals).sum(1)                         # DeepFM.py:127
        out["y_w"] = y_w
    emb = p["emb"][ids] * vals.reshape(B, F, 1)                 # DeepFM.py:130-132
    out["e"] = emb
    drop = list(cfg.dropout)

    def mlp(x, first_drop_index=0):
        for i, _h in enumerate(cfg.deep_layers):                # DeepFM.py:152-162
            x = _fc(x, p[f"mlp{i}/weights"], p[f"mlp{i}/biases"])
            if cfg.batch_norm:
                x, mm, mv = _bn(x, p, i, cfg, train)
                new_stats[f"bn_{i}/moving_mean"], new_stats[f"bn_{i}/moving_variance"] = mm, mv
            x = _dropout(x, drop[i] if i < len(drop) else 1.0, train, masks, f"mlp{i}")
        return x

    if cfg.model == "deepfm":
        sum_square = emb.sum(1) ** 2                            # DeepFM.py:133
        square_sum = (emb ** 2).sum(1)                          # DeepFM.py:134
        y_v = 0.5 * (sum_square - square_sum).sum(1)            # DeepFM.py:135
        out["y_v"] = y_v
        h = mlp(emb.reshape(B, F * K))                          # DeepFM.py:151
        y_d = _fc(h, p["deep_out/weights"], p["deep_out/biases"], relu=False).reshape(-1)
        y = p["bias"] * torch.ones_like(y_d) + y_w + y_v + y_d  # DeepFM.py:174-175
    elif cfg.model in ("fnn", "ipnn", "opnn"):
        flat = emb.reshape(B, F * K)
        if cfg.model == "fnn":                                  # PNN.py:139-140
            x = flat
        else:
            row, col = pair_index(F)
            pp = emb[:, row, :]                                 # PNN.py:148
            qq = emb[:, col, :]                                 # PNN.py:149
            if cfg.model == "ipnn":
                inner = (pp * qq).sum(-1)                       # PNN.py:152
                out["inner"] = inner
                x = torch.cat([flat, inner], 1)                 # PNN.py:153
            else:
                outer = torch.einsum("api,apj->apij", pp, qq).reshape(B, -1)   # PNN.py:166
                x = torch.cat([flat, outer], 1)                 # PNN.py:167
        h = mlp(x)
        y_d = _fc(h, p["deep_out/weights"], p["deep_out/biases"], relu=False).reshape(-1)
        y = p["bias"] * torch.ones_like(y_d) + y_w + y_d        # PNN.py:191-192
    elif cfg.model == "nfm":
        bi = 0.5 * (emb.sum(1) ** 2 - (emb ** 2).sum(1))        # NFM.py:126-128
        out["bi"] = bi
        x = _dropout(bi, drop[0], train, masks, "bi")           # NFM.py:136-137
        h = mlp(x)
        y_d = _fc(h, p["deep_out/weights"], p["deep_out/biases"], relu=False).reshape(-1)
        y = p["bias"] * torch.ones_like(y_d) + y_w + y_d        # NFM.py:153-154
    elif cfg.model == "afm":
        row, col = pair_index(F)
        ewp = emb[:, row, :] * emb[:, col, :]                   # AFM.py:134-138  [B,P,K]
        P = cfg.num_pairs
        a = ewp.reshape(-1, K)                                  # AFM.py:142
        for i, _a in enumerate(cfg.attention_layers):           # AFM.py:143-145
            a = _fc(a, p[f"att_mlp{i}/weights"], p[f"att_mlp{i}/biases"])
        aij = _fc(a, p["attention_out/weights"], p["attention_out/biases"], relu=False)  # AFM.py:147
        soft = torch.softmax(aij.reshape(B, P, 1), dim=1)       # AFM.py:151
        out["att"] = soft.reshape(B, P)
        soft = _dropout(soft, drop[0], train, masks, "att")     # AFM.py:152-153
        y_emb = (soft * ewp).sum(1)                             # AFM.py:156
        y_emb = _dropout(y_emb, drop[1] if len(drop) > 1 else 1.0, train, masks, "y_emb")  # AFM.py:158
        y_deep = _fc(y_emb, p["deep_out/weights"], p["deep_out/biases"], relu=False).reshape(-1)
        y = p["bias"] * torch.ones_like(y_deep) + y_w + y_deep  # AFM.py:166-167
    elif cfg.model == "dcn":
        x0 = emb.reshape(B, F * K)                              # DCN.py:138
        xl = x0
        for l in range(cfg.cross_layers):                       # DCN.py:141-145
            wl = p["cross_w"][l].reshape(-1, 1)
            xlw = xl @ wl
            xl = x0 * xlw + xl + p["cross_b"][l]
        out["x_cross"] = xl
        h = mlp(x0)                                             # DCN.py:161-176
        stack = torch.cat([xl, h], 1)                           # DCN.py:179
        y = _fc(stack, p["out_layer/weights"], p["out_layer/biases"], relu=False).reshape(-1)
    elif cfg.model == "mvm":
        all_order = emb + p["mvm_b"]                            # DeepMVM.py:145
        x_mvm = all_order[:, 0, :]                              # DeepMVM.py:146
        for i in range(1, F):                                   # DeepMVM.py:147-148
            x_mvm = x_mvm * all_order[:, i, :]
        out["x_mvm"] = x_mvm
        h = mlp(emb.reshape(B, F * K))                          # DeepMVM.py:167-181
        stack = torch.cat([x_mvm, h], 1)                        # DeepMVM.py:185
        y = _fc(stack, p["deep_out/weights"], p["deep_out/biases"], relu=False).reshape(-1)
    else:
        raise ValueError(cfg.model)
    out["y"] = y
    out["prob"] = torch.sigmoid(y)                              # DeepFM.py:176
    out["_new_stats"] = new_stats                               # type: ignore[assignment]
    return out


def l2_loss(t):
    """nn.l2_loss = sum(t**2)/2 [TF-1.4]."""
    return 0.5 * (t ** 2).sum()


def regularized_tables(cfg: Config) -> List[str]:
    """Variables that enter the loss through l2_loss: DeepFM.py:189-190, PNN.py:207,
    NFM.py:169, AFM.py:181 (linear + emb); DCN.py:199 (cross_b, cross_w, emb).  The MLP's
    weights_regularizer is dead code in the reference."""
    if cfg.model == "mvm":
        return ["emb", "mvm_b"]                                  # DeepMVM.py:197-199
    return ["cross_b", "cross_w", "emb"] if cfg.model == "dcn" else ["linear", "emb"]


def loss_fn(cfg: Config, p, y, labels):
    """mean sigmoid_cross_entropy_with_logits + l2 terms (DeepFM.py:188-190).
    xent(x,z) = max(x,0) - x z + log1p(exp(-|x|)) [TF-1.4]."""
    z = torch.as_tensor(labels).to(y.dtype).reshape(-1)
    xent = torch.clamp(y, min=0) - y * z + torch.log1p(torch.exp(-y.abs()))
    loss = xent.mean()
    for n in regularized_tables(cfg):
        loss = loss + cfg.l2_reg * l2_loss(p[n])
    return loss


# --------------------------------------------------------------------------------------
# backward (autograd on the restated forward) + TF-1.4 optimizers, dense over every variable
# --------------------------------------------------------------------------------------
def trainable(cfg: Config, p) -> List[str]:
    return [n for n in p if not n.endswith(NON_TRAINABLE_SUFFIXES)]


def grads(cfg: Config, p, ids, vals, labels, train=True, masks=None):
    """Dense gradients of the loss w.r.t. every trainable variable -- table gradients are
    densified exactly like TF does when an IndexedSlices gradient meets the dense l2_loss
    gradient."""
    names = trainable(cfg, p)
    leaves = {n: p[n].detach().clone().requires_grad_(True) for n in names}
    q = dict(p)
    q.update(leaves)
    out = forward(cfg, q, ids, vals, train=train, masks=masks)
    loss = loss_fn(cfg, q, out["y"], labels)
    gs = torch.autograd.grad(loss, [leaves[n] for n in names])
    return loss.detach(), {n: g for n, g in zip(names, gs)}, out


class Optimizer:
    """TF-1.4 update rules, dense.  DeepFM.py:204-211."""

    def __init__(self, cfg: Config, p):
        self.cfg = cfg
        self.kind = cfg.optimizer
        self.t = 0
        self.slots: Dict[str, Dict[str, torch.Tensor]] = {}
        for n in trainable(cfg, p):
            z = torch.zeros_like(p[n])
            if self.kind == "Adam":
                self.slots[n] = {"m": z.clone(), "v": z.clone()}
            elif self.kind == "Adagrad":
                self.slots[n] = {"accum": torch.full_like(p[n], 1e-8)}        # DeepFM.py:207
            elif self.kind == "Momentum":
                self.slots[n] = {"accum": z.clone()}
            elif self.kind == "ftrl":
                self.slots[n] = {"accum": torch.full_like(p[n], 0.1), "linear": z.clone()}
            else:
                raise NameError("optimizer")   # '--optimizer=GD' leaves `optimizer` unbound (DeepFM.py:204-213)

    def step(self, p, g):
        lr = self.cfg.learning_rate
        self.t += 1
        t = self.t
        for n, gn in g.items():
            s = self.slots[n]
            th = p[n]
            if self.kind == "Adam":                                            # DeepFM.py:205
                b1, b2, eps = 0.9, 0.999, 1e-8
                lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
                s["m"] = b1 * s["m"] + (1 - b1) * gn
                s["v"] = b2 * s["v"] + (1 - b2) * gn * gn
                p[n] = th - lr_t * s["m"] / (torch.sqrt(s["v"]) + eps)
            elif self.kind == "Adagrad":
                s["accum"] = s["accum"] + gn * gn
                p[n] = th - lr * gn / torch.sqrt(s["accum"])
            elif self.kind == "Momentum":                                      # DeepFM.py:209
                s["accum"] = 0.95 * s["accum"] + gn
                p[n] = th - lr * s["accum"]
            elif self.kind == "ftrl":                                          # DeepFM.py:211
                new_acc = s["accum"] + gn * gn
                sigma = (torch.sqrt(new_acc) - torch.sqrt(s["accum"])) / lr
                s["linear"] = s["linear"] + gn - sigma * th
                quad = torch.sqrt(new_acc) / lr
                p[n] = -s["linear"] / quad
                s["accum"] = new_acc
        return p


def train_step(cfg: Config, p, opt: Optimizer, ids, vals, labels, masks=None):
    loss, g, out = grads(cfg, p, ids, vals, labels, train=True, masks=masks)
    for k, v in out["_new_stats"].items():
        p[k] = v
    opt.step(p, g)
    return float(loss), out


# --------------------------------------------------------------------------------------
# tf.metrics.auc  (DeepFM.py:193-195) [TF-1.4]
# --------------------------------------------------------------------------------------
class StreamingAUC:
    def __init__(self, num_thresholds: int = 200):
        n = num_thresholds
        eps = 1e-7
        thr = [(i + 1) * 1.0 / (n - 1) for i in range(n - 2)]
        self.thr = np.array([0.0 - eps] + thr + [1.0 + eps], dtype=np.float32)
        self.tp = np.zeros(n, dtype=np.float64)
        self.fn = np.zeros(n, dtype=np.float64)
        self.tn = np.zeros(n, dtype=np.float64)
        self.fp = np.zeros(n, dtype=np.float64)

    def update(self, labels, pred):
        lab = np.asarray(labels).reshape(-1).astype(bool)
        pr = np.asarray(pred, dtype=np.float32).reshape(-1)
        gt = pr[None, :] > self.thr[:, None]
        self.tp += (gt & lab[None, :]).sum(1)
        self.fn += (~gt & lab[None, :]).sum(1)
        self.fp += (gt & ~lab[None, :]).sum(1)
        self.tn += (~gt & ~lab[None, :]).sum(1)

    def result(self) -> float:
        eps = np.float32(1e-6)
        tp, fn, tn, fp = (a.astype(np.float32) for a in (self.tp, self.fn, self.tn, self.fp))
        tpr = (tp + eps) / (tp + fn + eps)
        fpr = fp / (fp + tn + eps)
        n = len(tp)
        return float(np.sum((fpr[:n - 1] - fpr[1:]) * (tpr[:n - 1] + tpr[1:]) / np.float32(2.0)))


# --------------------------------------------------------------------------------------
# libsvm text -> (ids, vals, labels)   decode_libsvm, DeepFM.py:65-81
# --------------------------------------------------------------------------------------
_libc = ctypes.CDLL(None)
_libc.strtof.restype = ctypes.c_float
_libc.strtof.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p)]


def _strtof(tok: str) -> np.float32:
    """string_to_number(float32): correctly rounded decimal->binary32 [TF-1.4].  glibc
    strtof, NOT np.float32(float(tok)) which rounds twice."""
    b = tok.encode()
    end = ctypes.c_char_p()
    v = _libc.strtof(b, ctypes.byref(end))
    consumed = ctypes.cast(end, ctypes.c_void_p).value - ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p).value \
        if end.value is not None else len(b)
    if len(b) == 0 or (end.value is not None and len(end.value) != 0):
        raise ValueError("StringToNumberOp could not correctly convert string: %s" % tok)
    del consumed
    return np.float32(v)


def _strtoi32(tok: str) -> np.int32:
    s = tok
    if s.startswith(("+", "-")):
        body = s[1:]
    else:
        body = s
    if not body.isdigit():
        raise ValueError("StringToNumberOp could not correctly convert string: %s" % tok)
    v = int(s)
    if not (-2 ** 31 <= v < 2 ** 31):
        raise ValueError("StringToNumberOp could not correctly convert string: %s" % tok)
    return np.int32(v)


def parse_libsvm(text: str, field_size: int):
    """string_split(' ') drops empty tokens (skip_empty=True); token0 -> label f32; the rest
    split on ':' into [F,2]; col0 -> int32, col1 -> float32 (DeepFM.py:69-75).  A line whose
    id:val token count != field_size cannot be batched/reshaped (DeepFM.py:92,120-122)."""
    ids, vals, labels = [], [], []
    for ln, line in enumerate(text.split("\n")):
        if line == "":
            continue
        toks = [t for t in line.split(" ") if t != ""]
        if not toks:
            continue
        labels.append(_strtof(toks[0]))
        row_i, row_v = [], []
        for t in toks[1:]:
            parts = [x for x in t.split(":") if x != ""]
            if len(parts) != 2:
                raise ValueError("line %d: token %r is not id:val" % (ln, t))
            row_i.append(_strtoi32(parts[0]))
            row_v.append(_strtof(parts[1]))
        if len(row_i) != field_size:
            raise ValueError("line %d: %d id:val tokens, expected field_size=%d" % (ln, len(row_i), field_size))
        ids.append(row_i)
        vals.append(row_v)
    return (np.array(ids, dtype=np.int32).reshape(-1, field_size),
            np.array(vals, dtype=np.float32).reshape(-1, field_size),
            np.array(labels, dtype=np.float32))


# synthetic Criteo-shaped batches are generated by the product-side tf_repos_amd.synth so that the
# tests, bench.py and the CPU baseline all see identical inputs (the checker may import the product, never the reverse).
from tf_repos_amd.synth import synth_batch, to_libsvm  # noqa: E402,F401
