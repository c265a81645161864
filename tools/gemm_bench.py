"""Micro-benchmark of the fp32-MFMA GEMM entry points (run on the GPU box): python tools/gemm_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_repos_amd import capi

def bench(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

def main():
    L = capi.lib(); dev = torch.device("cuda:0"); st = capi.current_stream()
    for (M, K, N) in [(4096, 624, 400), (4096, 400, 400), (8192, 1248, 256), (4096, 1024, 1024), (16384, 624, 400)]:
        x = torch.randn(M, K, device=dev); w = torch.randn(K, N, device=dev) * 0.05; b = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev); dy = torch.randn(M, N, device=dev); dx = torch.empty(M, K, device=dev)
        dw = torch.empty(K, N, device=dev); db = torch.empty(N, device=dev); ws = torch.empty(64 * (K * N + N), device=dev)
        fl = 2.0 * M * K * N
        t = bench(lambda: capi.check(L.dctr_fc_fwd(capi.ptr(x), K, capi.ptr(w), capi.ptr(b), capi.ptr(y), N, M, K, N, 1, 1.0, 0, st)))
        ref = torch.relu(x @ w + b); err = (y - ref).abs().max().item()
        t2 = bench(lambda: capi.check(L.dctr_fc_bwd_data(capi.ptr(dy), N, capi.ptr(w), capi.ptr(dx), K, M, K, N, None, 0, 1.0, st)))
        t3 = bench(lambda: capi.check(L.dctr_fc_bwd_weights(capi.ptr(x), K, capi.ptr(dy), N, capi.ptr(dw), capi.ptr(db), M, K, N, capi.ptr(ws), ws.numel() * 4, st)))
        t4 = bench(lambda: torch.mm(x, w))
        print("M=%d K=%d N=%d  fwd %.1f us (%.1f TF) err %.2e | dgrad %.1f us (%.1f TF) | wgrad %.1f us (%.1f TF) | torch.mm %.1f us (%.1f TF)" % (
            M, K, N, t, fl / t / 1e6, err, t2, fl / t2 / 1e6, t3, fl / t3 / 1e6, t4, fl / t4 / 1e6), flush=True)

if __name__ == "__main__":
    main()
