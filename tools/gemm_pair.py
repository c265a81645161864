"""Do two GEMMs enqueued side by side (a layer's dgrad and the weight gradient of the layer above, as in the step) overlap?
Times N launches of each alone, then N of each on two streams at once, through the op-level C ABI.
    python tools/gemm_pair.py            (DCTR_LIB_VARIANT=w2 for the two-blocks-per-CU build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_repos_amd import capi

L = capi.lib(); dev = torch.device("cuda:0")
N_IT = 60


def run(fns, streams):
    """fns[i] launched N_IT times on streams[i]; all streams start together; -> us per (one launch of each)"""
    start = torch.cuda.Event(enable_timing=True)
    ends = [torch.cuda.Event(enable_timing=True) for _ in fns]
    for rep in range(2):
        torch.cuda.synchronize()
        start.record(torch.cuda.current_stream())
        for s in streams:
            s.wait_event(start)
        for i in range(N_IT):
            for fn, s in zip(fns, streams):
                fn(capi.C.c_void_p(s.cuda_stream))
        for e, s in zip(ends, streams):
            e.record(s)
        torch.cuda.synchronize()
    return max(start.elapsed_time(e) for e in ends) / N_IT * 1e3


def main():
    s1, s2 = torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=-1)
    for (M, K, N) in [(4096, 624, 400), (4096, 400, 400)]:
        x = torch.randn(M, K, device=dev); w = torch.randn(K, N, device=dev) * 0.05
        y = torch.randn(M, N, device=dev).relu(); dy = torch.randn(M, N, device=dev); dx = torch.empty(M, K, device=dev)
        dw = torch.empty(K, N, device=dev); db = torch.empty(N, device=dev); ws = torch.empty(64 * (K * N + N), device=dev)
        x2 = torch.randn(M, 400, device=dev); dw2 = torch.empty(400, N, device=dev)
        dgrad = lambda st: capi.check(L.dctr_fc_bwd_data(capi.ptr(dy), N, capi.ptr(w), capi.ptr(dx), K, M, K, N, None, 0, 1.0, st))
        wgrad = lambda st: capi.check(L.dctr_fc_bwd_weights(capi.ptr(x2), 400, capi.ptr(dy), N, capi.ptr(dw2), capi.ptr(db), M, 400, N, capi.ptr(ws), ws.numel() * 4, st))
        fwd = lambda st: capi.check(L.dctr_fc_fwd(capi.ptr(x), K, capi.ptr(w), capi.ptr(db), capi.ptr(y), N, M, K, N, 1, 0.5, 7, st))
        a, b, f = run([dgrad], [s1]), run([wgrad], [s1]), run([fwd], [s1])
        both = run([dgrad, wgrad], [s1, s2])
        two_d = run([dgrad, dgrad], [s1, s2])
        print("%s  dgrad %dx%dx%d %.1f us | wgrad 4096x400x%d (+ partial sum) %.1f us | fwd %.1f us | dgrad + wgrad side by side %.1f us (sum alone %.1f) | dgrad + dgrad %.1f (2x alone %.1f)" % (
            os.environ.get("DCTR_LIB_VARIANT", "base"), M, K, N, a, N, b, f, both, a + b, two_d, 2 * a), flush=True)


if __name__ == "__main__":
    import ctypes
    capi.C = ctypes
    main()
