"""Where the error of d att_mlp0/weights sits in tests/test_engine_gpu.py::test_afm_interaction_ops_through_the_c_abi[256-256-128-39-keep4]: in the columns
whose pre-activation z is within fp32 rounding of zero for some pair row (a ReLU decision that can fall either way), nowhere else.
usage (GPU box): python tools/afm_relu_flip_diag.py"""
import numpy as np, torch, sys
sys.path.insert(0, "/root/repo")
from tf_repos_amd import capi
from tf_repos_amd.engine import Engine, EngineConfig
dev = torch.device("cuda:0")
K, A, B, F, keep = 256, 256, 128, 39, (0.5, 0.5)
P = F * (F - 1) // 2
g = torch.Generator().manual_seed(11)
e = torch.randn(B, F, K, generator=g) * 0.3; W = torch.randn(K, A, generator=g) * 0.2; b = torch.randn(A, generator=g) * 0.1
wo = torch.randn(A, 1, generator=g) * 0.3; bo = torch.randn(1, generator=g) * 0.1; dy = torch.randn(B, K, generator=g) * 0.1
eng = Engine(EngineConfig(model="afm", field_size=F, feature_size=100, embedding_size=K, deep_layers=(1,), attention_layers=(A,), dropout=keep, l2_reg=0.0, learning_rate=1e-3, optimizer="Adam", max_batch=B, seed=5))
for name, v in (("att_mlp0/weights", W), ("att_mlp0/biases", b), ("attention_out/weights", wo), ("attention_out/biases", bo)): eng.set_param(name, v.numpy())
m_att = torch.from_numpy(eng.dropout_mask(capi.SITE_AFM_ATT, (B, P, 1), keep[0], step=0).astype(np.float64)); m_emb = torch.from_numpy(eng.dropout_mask(capi.SITE_AFM_YEMB, (B, K), keep[1], step=0).astype(np.float64))
e64 = e.double().requires_grad_(True); prm = [t.double().requires_grad_(True) for t in (W, b, wo, bo)]
row = [i for i in range(F - 1) for _ in range(i + 1, F)]; col = [j for i in range(F - 1) for j in range(i + 1, F)]
pp = e64[:, row, :] * e64[:, col, :]; z = pp.reshape(-1, K) @ prm[0] + prm[1]; ah = torch.relu(z); sc = (ah @ prm[2] + prm[3]).reshape(B, P, 1); sc.retain_grad()
soft = torch.softmax(sc, dim=1); y = ((soft * m_att / keep[0]) * pp).sum(1) * m_emb / keep[1]; y.backward(dy.double())
eng.afm_fwd(e.reshape(B, F * K).to(dev), train=True, want_att=True); eng.afm_bwd(dy.to(dev))
got = eng.get_grad("att_mlp0/weights").astype(np.float64).reshape(K, A); err = np.abs(got - prm[0].grad.numpy())
dsc = sc.grad.reshape(-1, 1).abs()
for win in (1e-6, 1e-7, 3e-8):
    near = (z.detach().abs() < win).double()
    slack = (((pp.detach().reshape(-1, K).abs() * dsc).t() @ near) * prm[2].detach().abs().reshape(1, -1)).numpy()
    bad = err > 1e-7 + slack
    print("window %.0e: %d near-zero pre-activations; columns with any: %d; elements over 1e-7 + budget: %d; max err %.2e, in columns with a near-zero z: %.2e, elsewhere %.2e" % (win, int(near.sum()), int((near.sum(0) > 0).sum()), int(bad.sum()), err.max(), err[:, near.sum(0).numpy() > 0].max() if near.sum() else 0, err[:, near.sum(0).numpy() == 0].max()))
