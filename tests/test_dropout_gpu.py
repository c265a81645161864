"""GPU parity of TRAINING WITH DROPOUT (keep_prob < 1) -- the configuration every reference operating point uses
(DeepFM.py:161-162, NFM.py:136-137, AFM.py:152-153,157-158; run.sh:13-22: 0.8 / 0.5; BASELINE c2: 0.5,0.5,0.5) and the
one bench.py times.

The engine's dropout is a pure function of (seed, global_step, site, element index) (include/deepctr_hip.h "dropout sites");
`Engine.dropout_mask` evaluates that function on the HOST through the C ABI (dctr_dropout_mask), the masks are handed to the
oracle (`masks=`), and both sides must then agree like at keep = 1: loss 1e-5 relative, every variable after the Adam steps
<= 5e-6 absolute.  A forward scale, a backward mask * 1/keep, an index layout (row * ld instead of row * width) or a site that
is off by a factor shows up here in every variable below it."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import deepctr_oracle as O
from oracle import multihot_oracle as M
from tests.util import dev_batch, make_pair
from tf_repos_amd import capi
from tf_repos_amd.engine import Engine, EngineConfig

pytestmark = pytest.mark.gpu

TOL = 5e-6


def oracle_masks(eng, ocfg, B):
    """{oracle key: 0/1 tensor} for the engine's NEXT train step"""
    F, K = ocfg.field_size, ocfg.embedding_size
    P = F * (F - 1) // 2
    keep = list(ocfg.dropout)
    m = {}
    if ocfg.model == "afm":                                      # AFM.py:152-153 (softmax weights [B,P,1]), :157-158 (y_emb [B,K])
        m["att"] = eng.dropout_mask(capi.SITE_AFM_ATT, (B, P, 1), keep[0])
        m["y_emb"] = eng.dropout_mask(capi.SITE_AFM_YEMB, (B, K), keep[1])
    else:
        if ocfg.model == "nfm":                                  # NFM.py:136-137
            m["bi"] = eng.dropout_mask(capi.SITE_NFM_BI, (B, K), keep[0])
        for i, h in enumerate(ocfg.deep_layers):                 # DeepFM.py:161-162
            m["mlp%d" % i] = eng.dropout_mask(capi.SITE_MLP(i), (B, h), keep[i])
    return {k: torch.from_numpy(v.astype(np.float32)) for k, v in m.items()}


def run_steps(ocfg, params, eng, dev, B, F, V, steps, seed0, loss_tol=1e-5, tol=TOL, t0=0):
    oopt = O.Optimizer(ocfg, params)
    oopt.t = t0                                                  # (Adam's bias correction follows global_step in the engine)
    for step in range(steps):
        ids, vals, labels = O.synth_batch(B, F, V, seed=seed0 + step)
        masks = oracle_masks(eng, ocfg, B)
        for k, v in masks.items():                               # the masks do drop and do keep
            assert 0.0 < float(v.mean()) < 1.0, k
        ref_loss, _ = O.train_step(ocfg, params, oopt, ids, vals, labels, masks=masks)
        loss = eng.train_step(*dev_batch(ids, vals, labels, dev))
        assert abs(loss - ref_loss) <= loss_tol * max(1.0, abs(ref_loss)), (step, loss, ref_loss)
    got = eng.get_params()
    for name, ref in params.items():
        diff = np.abs(got[name] - ref.numpy()).max()
        assert diff <= tol, (name, diff)


CASES = {
    # model: (F, V, B, K, layers, keep)
    "deepfm": (39, 3000, 128, 8, (48, 32, 16), (0.5, 0.5, 0.5)),          # README.md:49 --dropout=0.5,0.5,0.5
    "fnn": (39, 3000, 96, 8, (32, 16), (0.8, 0.6)),
    "ipnn": (39, 3000, 96, 8, (32, 16), (0.5, 0.5)),                       # run.sh:15
    "nfm": (39, 3000, 128, 16, (32, 16), (0.5, 0.8)),                      # NFM.py:136-137: keep[0] drops the bi-interaction AND layer 0
    "dcn": (39, 3000, 128, 8, (32, 16), (0.8, 0.8)),
    "mvm": (39, 3000, 96, 8, (32, 16), (0.8, 0.5)),
    "afm": (12, 800, 64, 8, (32, 16), (0.7, 0.6)),                         # AFM.py:152-158: attention / pooled-embedding dropout
}


@pytest.mark.parametrize("use_graph", [True, False])
@pytest.mark.parametrize("model", list(CASES))
def test_train_with_dropout_matches_oracle(model, use_graph, dev):
    F, V, B, K, layers, keep = CASES[model]
    ocfg, params, eng = make_pair(model, B=B, F=F, V=V, K=K, layers=layers, keep=keep, seed=1234, use_graph=use_graph, lr=1e-2, l2=1e-3)
    run_steps(ocfg, params, eng, dev, B, F, V, steps=3, seed0=700)
    assert eng.global_step == 3
    eng.close()


@pytest.mark.parametrize("K,H", [(16, 256), (32, 64)])
def test_outer_pnn_fused_first_layer_with_dropout(K, H, dev):
    """opnn_finish_kernel applies bias / ReLU / dropout of the fused Outer-PNN first layer (gemm_dr.hip)"""
    F, V, B = 10, 700, 70
    ocfg, params, eng = make_pair("opnn", B=B, F=F, V=V, K=K, layers=(H, 32), keep=(0.5, 0.8), seed=77, lr=1e-3, l2=1e-4)
    run_steps(ocfg, params, eng, dev, B, F, V, steps=2, seed0=710)
    eng.close()


@pytest.mark.parametrize("model", ["deepfm", "nfm"])
def test_batch_norm_then_dropout(model, dev):
    """relu -> batch_norm -> dropout (DeepFM.py:156-162): the mask rides in bn_apply / bn_backward instead of the GEMM epilogue.
    (Momentum as in the other BN tests: the bias gradients are exactly zero in exact arithmetic, Adam amplifies their rounding.)"""
    F, V, B, K = 39, 3000, 128, 8
    ocfg, params, eng = make_pair(model, B=B, F=F, V=V, K=K, layers=(32, 16), keep=(0.5, 0.8), seed=5, opt="Momentum", lr=1e-2,
                                  batch_norm=True)
    run_steps(ocfg, params, eng, dev, B, F, V, steps=3, seed0=720, loss_tol=2e-5, tol=2e-5)
    eng.close()


@pytest.mark.parametrize("opt", ["Adagrad", "Momentum", "ftrl"])
def test_dropout_with_the_other_optimizers(opt, dev):
    F, V, B, K = 39, 3000, 128, 8
    lr = {"Adagrad": 1e-2, "Momentum": 1e-2, "ftrl": 5e-2}[opt]
    ocfg, params, eng = make_pair("deepfm", B=B, F=F, V=V, K=K, layers=(32, 16), keep=(0.5, 0.5), seed=9, opt=opt, lr=lr)
    run_steps(ocfg, params, eng, dev, B, F, V, steps=3, seed0=730)
    eng.close()


def test_ragged_last_batch_and_restored_global_step(dev):
    """the mask index is row * width whatever max_batch is, and the step number comes from the restored global_step"""
    F, V, K = 39, 3000, 8
    ocfg, params, eng = make_pair("deepfm", B=77, F=F, V=V, K=K, layers=(40, 24), keep=(0.5, 0.5), seed=31, max_batch=256)
    eng.global_step = 41
    run_steps(ocfg, params, eng, dev, 77, F, V, steps=2, seed0=740, t0=41)
    assert eng.global_step == 43
    eng.close()


def _c2_step(gemm):
    """BASELINE c2 at full size with its own dropout (0.5, 0.5, 0.5): one dense-exact Adam step against the oracle"""
    dev = torch.device("cuda:0")
    F, V, B, K = 39, 1_000_000, 4096, 16
    ocfg, params, eng = make_pair("deepfm", B=B, F=F, V=V, K=K, layers=(400, 400, 400), keep=(0.5, 0.5, 0.5), seed=20260924, l2=1e-4,
                                  lr=5e-4, scale=0.01, use_graph=False)
    plan = " ".join(capi_plan(op, B, k, 400) for op, k in (("f", F * K), ("d", 400), ("w", 400)))     # (M rows, K in, N out)
    assert ("dr" in plan) == (gemm != "lds"), plan
    run_steps(ocfg, params, eng, dev, B, F, V, steps=1, seed0=20260924, tol=2e-6)
    eng.close()
    print("c2 dropout step ok:", gemm, plan)


def capi_plan(op, m, k, n):
    import ctypes as C
    buf = C.create_string_buffer(128)
    capi.check(capi.lib().dctr_gemm_plan(op.encode(), m, k, n, buf, 128))
    return buf.value.decode()


@pytest.mark.parametrize("gemm", ["fdw", "lds"])
def test_c2_full_size_step_with_dropout(gemm, dev):
    """both GEMM families carry the mask in their epilogues (gemm_dr.h direct-to-register, gemm.hip LDS-tiled): the family is a
    process-wide choice (DCTR_GEMM), so each runs in a child process"""
    env = dict(os.environ, DCTR_GEMM=gemm)
    r = subprocess.run([sys.executable, "-c", "import sys; from tests.test_dropout_gpu import _c2_step as f; f(sys.argv[1])", gemm],
                       env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "c2 dropout step ok" in r.stdout


# ---- CSR models: the towers of DIN / ESMM (DIN.py:205-206, DeepCvrMTL.py:179-180,200-201) and DIN's attention units (DIN.py:166) ----

def _mh_pair(model, B, keep, att=(), K=8, layers=(32, 16), opt="Adam", seed=21):
    Fc, V = 6, 800
    ocfg = M.Config(model=model, field_size=Fc, feature_size=V, embedding_size=K, deep_layers=layers, dropout=keep, l2_reg=1e-3,
                    learning_rate=1e-2, optimizer=opt, ctr_task_wgt=0.4, attention_layers=att)
    ecfg = EngineConfig(model=model, field_size=ocfg.n_slots, feature_size=V, embedding_size=K, deep_layers=layers, dropout=keep,
                        l2_reg=1e-3, learning_rate=1e-2, optimizer=opt, max_batch=B, max_entries=B * (ocfg.n_slots + 40), ctr_task_wgt=0.4,
                        attention_layers=att or (256,), att_pairs=[(Fc + i, Fc + 4 + i) for i in range(4)] if att else (), seed=seed)
    params = M.init_params(ocfg, seed=4, scale=0.2 if att else 0.05)
    eng = Engine(ecfg)
    eng.set_params(params)
    return ocfg, params, eng


def _mh_masks(eng, ocfg, batch, off):
    B = batch["feat_ids"].shape[0]
    keep = list(ocfg.dropout)
    m = {}
    prefixes = ["ctr_", "cvr_"] if ocfg.model == "esmm" else [""]
    for t, pre in enumerate(prefixes):
        for i, h in enumerate(ocfg.deep_layers):
            site = capi.SITE_MLP(i) if t == 0 else capi.SITE_MLP2(i)
            m["%smlp%d" % (pre, i)] = eng.dropout_mask(site, (B, h), keep[i])
    if ocfg.attention_layers:
        # the attention MLP runs over the nnz entry rows of the slot CSR (din_att.hip); unit u's rows are the entries of its slot
        S = ocfg.n_slots
        nnz = int(off[-1])
        slot_of = np.repeat(np.arange(B * S) % S, np.diff(off))
        for i, a in enumerate(ocfg.attention_layers):
            full = eng.dropout_mask(capi.SITE_MLP2(i), (nnz, a), keep[i])
            for j, u in enumerate(M.MULTI_W):
                m["%s/att_fc%d" % (u, i)] = full[slot_of == ocfg.field_size + j]
    return {k: torch.from_numpy(v.astype(np.float32)) for k, v in m.items()}


@pytest.mark.parametrize("model,att", [("din", ()), ("esmm", ()), ("din", (16, 8))])
def test_csr_models_train_with_dropout(model, att, dev):
    B = 64
    ocfg, params, eng = _mh_pair(model, B, keep=(0.5, 0.8), att=att)
    oopt = M.Optimizer(ocfg, params)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for step in range(3):
        batch = M.synth_batch(ocfg, B, seed=760 + step)
        off, ids, wts = M.slot_csr(ocfg, batch)
        masks = _mh_masks(eng, ocfg, batch, off)
        if model == "esmm":                                  # the two towers draw independently (two nn.dropout ops)
            assert not torch.equal(masks["ctr_mlp0"], masks["cvr_mlp0"])
        ref_loss, _ = M.train_step(ocfg, params, oopt, batch, masks=masks)
        loss = eng.train_step_csr(t(off), t(ids), t(wts), t(batch["y"]), t(batch["z"]) if model == "esmm" else None)
        assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (step, loss, ref_loss)
    got = eng.get_params()
    for name, ref in params.items():
        diff = np.abs(got[name] - ref.numpy()).max()
        assert diff <= TOL, (name, diff)
    eng.close()
