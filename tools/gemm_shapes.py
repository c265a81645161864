"""Times fwd / dgrad / wgrad of one dense layer for a list of shapes (hipEvents around 50 back-to-back launches).
usage (GPU box): python tools/gemm_shapes.py 4096x624x400 4096x640x448 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_repos_amd import capi
L = capi.lib(); dev = torch.device("cuda:0"); st = capi.current_stream()
shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[1:]] or [(4096, 624, 400)]


def timed(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, K, N in shapes:
    x = torch.randn(M, K, device=dev); w = torch.randn(K, N, device=dev) * 0.05; b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev); dy = torch.randn(M, N, device=dev); dx = torch.empty(M, K, device=dev)
    dw = torch.empty(K, N, device=dev); db = torch.empty(N, device=dev); ws = torch.empty(64 * (K * N + N), device=dev)
    fl = 2.0 * M * K * N
    tf = timed(lambda: capi.check(L.dctr_fc_fwd(capi.ptr(x), K, capi.ptr(w), capi.ptr(b), capi.ptr(y), N, M, K, N, 1, 0.5, 0, st)))
    td = timed(lambda: capi.check(L.dctr_fc_bwd_data(capi.ptr(dy), N, capi.ptr(w), capi.ptr(dx), K, M, K, N, capi.ptr(x), K, 0.5, st)))
    tw = timed(lambda: capi.check(L.dctr_fc_bwd_weights(capi.ptr(x), K, capi.ptr(dy), N, capi.ptr(dw), capi.ptr(db), M, K, N, capi.ptr(ws), ws.numel() * 4, st)))
    print("%dx%dx%d  fwd %.1f us (%.1f TF)  dgrad %.1f us (%.1f TF)  wgrad+reduce %.1f us (%.1f TF)" % (M, K, N, tf, fl / tf / 1e6, td, fl / td / 1e6, tw, fl / tw / 1e6), flush=True)
