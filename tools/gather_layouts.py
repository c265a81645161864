"""The gather (DeepFM.py:125-135: w[id], V[id] * val, FM sums) on three table layouts, HBM-resident and cache-resident, K = 16 and
K = 32 (c5's shape), uniform ids:
  separate : emb [V, K] + linear [V]                    (the engine's layout: variables keep the reference's shapes)
  rec128   : [row | weight | pad] records padded to whole 128-byte granules (K = 16: 32 floats, K = 32: 64 floats)
  rec_k4   : [row | weight | pad] records of K + 4 floats (what the row-sharded exchange ships)
Reports time per launch (back-to-back launches through the C ABI, hipEvents) and algorithmic TB/s (B (F (12 + 8K) + 8) bytes);
under `rocprofv3 --pmc FETCH_SIZE` (tools/prof_cmd.sh) the bytes the memory side moved.  The dense-exact sweep and the
touched-rows step are priced from the record sizes at the measured streaming rate (the table says how).
usage (GPU box): python tools/gather_layouts.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tf_repos_amd import capi

dev = torch.device("cuda", 0)
L = capi.lib()
B, F = 4096, 39
ITERS = int(os.environ.get("ITERS", "200"))


def run(K, V, layout):
    if layout == "separate":
        ld, base = K, torch.zeros(V * K, device=dev)
        lin, lin_ld = torch.zeros(V, device=dev), 1
        lin_ptr = capi.ptr(lin)
    else:
        ld = {16: 32, 32: 64}[K] if layout == "rec128" else K + 4
        base = torch.zeros(V * ld, device=dev)
        lin, lin_ld = None, ld
        lin_ptr = capi.ptr(base.data_ptr() + 4 * K)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    ids = [torch.randint(0, V, (B, F), device=dev, dtype=torch.int32, generator=g) for _ in range(8)]
    vals = torch.rand(B, F, device=dev)
    e = torch.empty(B, F * K, device=dev)
    yw, yv, S = torch.empty(B, device=dev), torch.empty(B, device=dev), torch.empty(B, K, device=dev)
    status = torch.zeros(4, dtype=torch.int32, device=dev)
    st = capi.current_stream()

    def launch(i):
        capi.check(L.dctr_embed_gather_strided(capi.ptr(base), ld, lin_ptr, lin_ld, V, capi.ptr(ids[i % 8]), capi.ptr(vals), B, F, K, 1, capi.ptr(e),
                                               F * K, capi.ptr(yw), capi.ptr(S), capi.ptr(yv), capi.ptr(status), st))
    for i in range(10):
        launch(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(ITERS):
        launch(i)
    e1.record()
    e1.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / ITERS
    alg = B * (F * (12 + 8 * K) + 8)
    row_bytes = ld * 4 if layout != "separate" else (K + 1) * 4
    del base, lin
    torch.cuda.empty_cache()
    return {"K": K, "V": V, "layout": layout, "row_floats": ld if layout != "separate" else K + 1, "table_GB": round(V * row_bytes / 1e9, 2),
            "gather_us": round(us, 2), "algorithmic_TBps": round(alg / us / 1e6, 3),
            # dense-exact Adam sweep: theta, m, v read + written = 6 streams of the record (priced at 5.3 TB/s, the c5-shape sweep's rate)
            "sweep_bytes_per_row": 6 * row_bytes, "sweep_us_per_M_rows_at_5.3TBps": round(6 * row_bytes * 1e6 / 5.3e12 * 1e6, 1)}


for K, V in ((16, 64 << 20), (32, 32 << 20), (16, 1_000_000), (32, 1_000_000)):
    for layout in ("separate", "rec128", "rec_k4"):
        print(json.dumps(run(K, V, layout)), flush=True)
