"""Runs the three GEMMs of one layer a few times (for rocprofv3 counter passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_repos_amd import capi
M, K, N = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 624, 400))]
L = capi.lib(); dev = torch.device("cuda:0"); st = capi.current_stream()
x = torch.randn(M, K, device=dev); w = torch.randn(K, N, device=dev) * 0.05; b = torch.randn(N, device=dev)
y = torch.empty(M, N, device=dev); dy = torch.randn(M, N, device=dev); dx = torch.empty(M, K, device=dev)
dw = torch.empty(K, N, device=dev); db = torch.empty(N, device=dev); ws = torch.empty(64 * (K * N + N), device=dev)
for _ in range(5):
    capi.check(L.dctr_fc_fwd(capi.ptr(x), K, capi.ptr(w), capi.ptr(b), capi.ptr(y), N, M, K, N, 1, 1.0, 0, st))
    capi.check(L.dctr_fc_bwd_data(capi.ptr(dy), N, capi.ptr(w), capi.ptr(dx), K, M, K, N, None, 0, 1.0, st))
    capi.check(L.dctr_fc_bwd_weights(capi.ptr(x), K, capi.ptr(dy), N, capi.ptr(dw), capi.ptr(db), M, K, N, capi.ptr(ws), ws.numel() * 4, st))
torch.cuda.synchronize()
