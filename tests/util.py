"""Shared helpers for the parity tests (tests may use the oracle; the product never does)."""
import numpy as np
import torch

from oracle import deepctr_oracle as O
from tf_repos_amd.engine import Engine, EngineConfig


def make_pair(model, B=64, F=39, V=2000, K=8, layers=(32, 16), cross=2, opt="Adam", l2=1e-3, lr=1e-2,
              table_mode="dense_exact", seed=0, keep=None, scale=0.05, use_graph=True, max_batch=None, att=(16,), batch_norm=False):
    keep = tuple(keep) if keep is not None else ((1.0, 1.0) if model == "afm" else tuple(1.0 for _ in layers))
    ocfg = O.Config(model=model, field_size=F, feature_size=V, embedding_size=K, deep_layers=tuple(layers), dropout=keep,
                    cross_layers=cross, l2_reg=l2, learning_rate=lr, optimizer=opt, attention_layers=tuple(att), batch_norm=batch_norm)
    ecfg = EngineConfig(model=model, field_size=F, feature_size=V, embedding_size=K, deep_layers=tuple(layers), dropout=keep,
                        cross_layers=cross, l2_reg=l2, learning_rate=lr, optimizer=opt, table_mode=table_mode, attention_layers=tuple(att),
                        max_batch=max_batch or B, seed=seed, use_graph=use_graph, batch_norm=batch_norm)
    params = O.init_params(ocfg, seed=seed + 1, scale=scale)
    eng = Engine(ecfg)
    eng.set_params(params)
    return ocfg, params, eng


def dev_batch(ids, vals, labels, dev):
    return (torch.from_numpy(np.ascontiguousarray(ids)).to(dev), torch.from_numpy(np.ascontiguousarray(vals)).to(dev),
            torch.from_numpy(np.ascontiguousarray(labels)).to(dev))
