#!/usr/bin/env python
"""bench.py -- examples/sec of the DeepFM training step (BASELINE.json metric) on N MI355X GPUs of one node.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workloads (--config):
  c2 (default; BASELINE.json configs[1], the configuration the metric is quoted on): DeepFM, 39 fields, vocab 1e6, emb_dim 16,
      batch 4096 per GPU (weak scaling), MLP 400-400-400 keep 0.5, Adam lr 5e-4, l2_reg 1e-4 (deep_ctr/README.md:49).
  c5 (BASELINE.json configs[4]): DeepFM vocab 1e8, emb_dim 32, batch 8192 per GPU (65 536 at 8 GPUs), table row-sharded.
f32 arithmetic, synthetic Criteo-shaped libsvm-equivalent tensors already resident in HBM, random-init N(0,0.01) weights.
A "step" = forward + loss + backward + optimizer over one batch, table optimizer in DENSE-EXACT mode (what the reference's TF
graph does: l2_loss on the tables makes the optimizer stream all V rows every step).
Prints ONE JSON line (rank 0).  `roofline` is the kernel TEMPLATE with the most time in the timed steps (the weight-gradient product of the MLP;
every product's dispatch carries its own start / stop events) with `roofline.family` = all nine MLP products; `cpu_baseline` is the torch-CPU restatement of the reference's TF-1.4 graph (oracle/, "port") timed on this
box's host cores by the protocol of BASELINE.md section 3.
"""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F32_PEAK_TFS = 157.3  # f32-input MFMA peak
MFMA_BF16_PEAK_TFS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md; AMD's 5 PF figure includes 2:1 sparsity)
# gemm_mode "split": one f32 multiply-add = six bf16 plane products on the bf16 pipe -> the matrix-pipe ceiling in f32-equivalent flops
MFMA_SPLIT_PEAK_TFS = MFMA_BF16_PEAK_TFS / 6.0
DTYPE = {"exact": "f32", "split": "f32 (3-plane bf16 split, 6 products, f32 accumulate)"}

CONFIGS = {
    "c2": dict(model="deepfm", field_size=39, feature_size=1_000_000, embedding_size=16, batch=4096,
               deep_layers=(400, 400, 400), dropout=(0.5, 0.5, 0.5), l2_reg=1e-4, learning_rate=5e-4, optimizer="Adam",
               name="DeepFM 39 fields, vocab 1e6, emb_dim 16, batch 4096/GPU, MLP 400-400-400 keep 0.5, Adam (BASELINE configs[1])"),
    "c5": dict(model="deepfm", field_size=39, feature_size=100_000_000, embedding_size=32, batch=8192,
               deep_layers=(400, 400, 400), dropout=(0.5, 0.5, 0.5), l2_reg=1e-4, learning_rate=5e-4, optimizer="Adam",
               name="DeepFM 39 fields, vocab 1e8 row-sharded, emb_dim 32, batch 8192/GPU (65536 at 8 GPUs), MLP 400-400-400 keep 0.5, "
                    "Adam (BASELINE configs[4])"),
}
PMC_FILE = os.path.join("profiles", "r06_pmc_traffic.txt")
STATS_FILE = os.path.join("profiles", "r06_kernel_stats.txt")     # rocprofv3 --kernel-trace --stats of this same command
LIB_FILE = os.path.join("tf_repos_amd", "_lib", "libdeepctr_hip.so")


def cpu_baseline(w, steps=200, warmup=20):
    """the oracle (torch-CPU fp32 restatement of the TF-1.4 graph, NOT TF) on this box's host cores --
    same model, same batch shape, table gradient densified, dense Adam over all rows.  20 warm-up + `steps` timed steps, MEDIAN
    step time; thread count = best of a sweep up to nproc; the text parse of the same batches (split ' ' then ':', 10 worker
    threads, DeepFM.py:65-92) timed separately, so an input-bound baseline is not mistaken for a compute-bound one.
    value = min(parse-only, compute-only): tf.data parses in background threads, the slower side sets the rate."""
    import numpy as np
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from oracle import deepctr_oracle as O
    cores = os.cpu_count() or 1
    cfg = O.Config(model=w["model"], field_size=w["field_size"], feature_size=w["feature_size"],
                   embedding_size=w["embedding_size"], deep_layers=w["deep_layers"], dropout=w["dropout"],
                   l2_reg=w["l2_reg"], learning_rate=w["learning_rate"], optimizer=w["optimizer"])
    p = O.init_params(cfg, seed=1, scale=0.01)
    opt = O.Optimizer(cfg, p)
    B, F = w["batch"], w["field_size"]
    batches = [O.synth_batch(B, F, cfg.feature_size, seed=20260924 + i) for i in range(4)]
    sweep, best_nt, best_t = {}, 1, float("inf")
    nts = sorted({min(cores, n) for n in (8, 16, 32, 64, 128, 256)} | {cores})
    for nt in nts:
        torch.set_num_threads(nt)
        O.train_step(cfg, p, opt, *batches[0])
        t0 = time.perf_counter()
        for i in range(3):
            O.train_step(cfg, p, opt, *batches[i % 4])
        t = (time.perf_counter() - t0) / 3
        sweep[nt] = round(B / t, 1)
        if t < best_t:
            best_nt, best_t = nt, t
        if t > 3 * best_t:           # far past the optimum: more threads only get slower on the memory-bound table passes
            break
    torch.set_num_threads(best_nt)
    for i in range(warmup):
        O.train_step(cfg, p, opt, *batches[i % 4])
    ts = []
    for i in range(steps):
        t0 = time.perf_counter()
        O.train_step(cfg, p, opt, *batches[i % 4])
        ts.append(time.perf_counter() - t0)
    compute = B / float(np.median(ts))
    # parse-only: the libsvm text of the same batches, 10 worker threads over line chunks
    text = [O.to_libsvm(*b) for b in batches[:2]]

    def parse_chunk(lines):
        lab = np.empty(len(lines), np.float32)
        ids = np.empty((len(lines), F), np.int32)
        vals = np.empty((len(lines), F), np.float32)
        for r, ln in enumerate(lines):
            tok = ln.split(" ")
            lab[r] = float(tok[0])
            kv = [t.split(":") for t in tok[1:]]
            ids[r] = [int(a) for a, _ in kv]
            vals[r] = [float(b) for _, b in kv]
        return ids, vals, lab

    pool = ThreadPoolExecutor(10)
    pt = []
    for rep in range(4):
        lines = text[rep % 2].rstrip("\n").split("\n")
        chunks = [lines[i::10] for i in range(10)]
        t0 = time.perf_counter()
        list(pool.map(parse_chunk, chunks))
        pt.append(time.perf_counter() - t0)
    pool.shutdown()
    parse = B / float(np.median(pt))
    return {"value": round(min(parse, compute), 1), "unit": "examples/sec", "cores": best_nt, "kind": "port",
            "compute_only": round(compute, 1), "parse_only": round(parse, 1), "thread_sweep_examples_per_sec": sweep,
            "sample": "torch-CPU fp32 restatement of the TF-1.4 reference graph (TF not installable): %d warm-up + %d timed train steps "
                      "of the same workload (batch %d), MEDIAN step; dense table gradient + dense Adam over all rows; %d threads = best "
                      "of the sweep %s on a %d-core host; parse-only = Python split/float parse of the same libsvm lines on 10 "
                      "worker threads; value = min(parse-only, compute-only)" % (warmup, steps, B, best_nt, list(sweep), cores)}


def pmc_traffic_bytes(kernel_name):
    """HBM bytes per launch of ONE kernel (exact demangled-name match, template arguments included) from the committed PMC
    summary (collected by tools/profile_round.sh in separate --pmc passes: a live bench run cannot sample counters).
    bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE counts half of the 16-B/lane streams; MI355X_MICROARCH.md
    HBM section).  None when the summary is absent or does not list the kernel."""
    path = os.path.join(ROOT, PMC_FILE)
    if not os.path.exists(path):
        return None
    for line in open(path):
        if line.startswith(kernel_name + "("):          # the name with its template arguments, then the argument list
            mf = re.search(r"FETCH_SIZE=([0-9.e+]+)", line)
            mw = re.search(r"WRITE_SIZE=([0-9.e+]+)", line)
            if mf and mw:
                return int((2 * float(mf.group(1)) + float(mw.group(1))) * 1024)
    return None


def profile_is_stale(rel):
    """a committed profile made from other sources than the library's present ones describes other kernels than the ones that just
    ran.  tools/profile_round.sh stamps its summaries with the sha256 of csrc/ + the header (tf_repos_amd.build.sources_hash); a summary
    without a stamp is judged by file times, as in round 3."""
    a, b = os.path.join(ROOT, rel), os.path.join(ROOT, LIB_FILE)
    if not os.path.exists(a):
        return False
    for line in open(a):
        if line.startswith("# sources sha256:"):
            from tf_repos_amd.build import sources_hash
            return line.split(":", 1)[1].strip() != sources_hash()
    return os.path.exists(b) and os.path.getmtime(a) < os.path.getmtime(b)


def rocprof_avg_us(kernel_name):
    """average duration (us) of one kernel in the committed rocprofv3 --stats summary of `python bench.py` (None if absent)"""
    path = os.path.join(ROOT, STATS_FILE)
    if not os.path.exists(path):
        return None
    for line in open(path):
        if line.startswith(kernel_name[:60]):
            f = line.split()
            try:
                return float(f[-2])
            except (ValueError, IndexError):
                return None
    return None


def rocprof_family(prefix):
    """(total microseconds, calls) per kernel template whose demangled name starts with `prefix`, from the committed rocprofv3 --stats
    summary of `python bench.py`: {template name: (total_us, calls)}; {} if the summary is absent"""
    path = os.path.join(ROOT, STATS_FILE)
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        if line.startswith(prefix):
            name = line[:line.index("(")] if "(" in line else line.split()[0]
            f = line.split()
            try:
                out[name] = (float(f[-3]), int(f[-4]))           # ... calls total_us avg_us pct
            except (ValueError, IndexError):
                pass
    return out


def fill_normal_(t, scale, seed):
    """In-place N(0, scale) on the device, in chunks (c5's 12.8 GB table never exists on the host)."""
    import torch
    g = torch.Generator(device=t.device)
    g.manual_seed(seed)
    if not t.is_contiguous():            # a table kept as row records (Engine.param_tensor): by blocks of rows
        rows = max(1, (1 << 26) // max(1, t[0].numel()))
        for s in range(0, t.shape[0], rows):
            t[s:s + rows].normal_(0.0, scale, generator=g)
        return
    flat = t.view(-1)
    step = 1 << 28
    for s in range(0, flat.numel(), step):
        flat[s:s + step].normal_(0.0, scale, generator=g)


def hbm_resident_gather(dev, K=16, V=64 * 1024 * 1024, B=4096, F=39, iters=200, zipf=False):
    """The gather kernel alone on a table far larger than the 256 MB Infinity Cache (V = 64 M rows x K = 16: 4.3 GB) with uniform
    ids -- or (zipf) the workload's own Criteo-shaped ids: 13 always-hit numeric ids, 26 categorical fields with field-disjoint Zipf
    ranks (tf_repos_amd/synth.py, the generator the timed steps use): algorithmic bytes B (F (12 + 8K) + 8) over the hipEvent time of
    back-to-back launches through the op-level C ABI."""
    import numpy as np
    import torch
    from tf_repos_amd import capi
    from tf_repos_amd.synth import synth_batch
    L = capi.lib()
    emb = torch.zeros(V, K, device=dev)
    lin = torch.zeros(V, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    nb = 8
    if zipf:
        ids = [torch.from_numpy(synth_batch(B, F, V, seed=20260924 + 500 + i)[0]).to(dev) for i in range(nb)]
    else:
        ids = [torch.randint(0, V, (B, F), device=dev, dtype=torch.int32, generator=g) for _ in range(nb)]
    vals = torch.rand(B, F, device=dev)
    e = torch.empty(B, F * K, device=dev)
    yw, yv, S = torch.empty(B, device=dev), torch.empty(B, device=dev), torch.empty(B, K, device=dev)
    status = torch.zeros(4, dtype=torch.int32, device=dev)
    st = capi.current_stream()

    def launch(i):
        capi.check(L.dctr_embed_gather_fwd(capi.ptr(emb), capi.ptr(lin), V, capi.ptr(ids[i % nb]), capi.ptr(vals), B, F, K, 1, capi.ptr(e), F * K,
                                           capi.ptr(yw), capi.ptr(S), capi.ptr(yv), capi.ptr(status), st))
    for i in range(10):
        launch(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        launch(i)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / iters
    del emb, lin
    torch.cuda.empty_cache()
    return ms, B * (F * (12 + 8 * K) + 8)


def end_to_end(w, epochs=40, lines=32768 * 12):
    """What a user of the reference runs: `tf.estimator.Estimator(model_fn, ...).train(input_fn)` over a libsvm TEXT file
    (examples/ctr_estimator.py: the model_fn / input_fn of DeepFM.py:63-221 written against the tensorflow surface of tf_shim).
    One short call first: it pays the first-use costs (library load, engine allocation) AND parses the file's text (C parser,
    threads), leaving the parsed rows in memory and in a .dctr.npz beside the file -- so the timed multi-epoch call measures batching,
    H2D through the engine's input slots, the train steps and the checkpoint save, not text parsing; `cold_text_one_epoch` is one
    epoch of a file seen for the first time, parse included.  examples/s = examples / wall time of the whole train() call.  Never
    `value`: reported product-level rates beside it."""
    import importlib.util
    import tempfile
    import torch
    from tf_repos_amd.synth import synth_batch, to_libsvm
    import tf_repos_amd.tf_shim as shim
    d = tempfile.mkdtemp(prefix="dctr_e2e_")
    B, F, V = w["batch"], w["field_size"], w["feature_size"]
    chunk = "".join(to_libsvm(*synth_batch(4096, F, V, seed=77 + i)) for i in range(8))        # 32 768 distinct lines
    path, small = os.path.join(d, "tr.libsvm"), os.path.join(d, "warm.libsvm")
    with open(path, "w") as f:
        for _ in range(max(1, lines // 32768)):
            f.write(chunk)
    with open(small, "w") as f:
        f.write(chunk)
    n_lines = 32768 * max(1, lines // 32768)
    shim.install()
    spec = importlib.util.spec_from_file_location("ctr_estimator_example", os.path.join(ROOT, "examples", "ctr_estimator.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    p = dict(model=w["model"], field_size=F, feature_size=V, embedding_size=w["embedding_size"], learning_rate=w["learning_rate"], l2_reg=w["l2_reg"],
             deep_layers=",".join(str(h) for h in w["deep_layers"]), dropout=",".join(str(k) for k in w["dropout"]), cross_layers=3,
             optimizer=w["optimizer"])
    est = mod.build_estimator(p, os.path.join(d, "ckpt"), log_steps=10 ** 9)
    est.train(input_fn=lambda: mod.input_fn([small], num_epochs=1, batch_size=B))
    torch.cuda.synchronize()

    def timed(ep):
        t0 = time.perf_counter()
        est.train(input_fn=lambda: mod.input_fn([path], num_epochs=ep, batch_size=B))
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    short = max(1, epochs // 10)
    dt_short, dt = timed(short), timed(epochs)
    steps, steps_short = n_lines * epochs // B, n_lines * short // B
    per_step = (dt - dt_short) / max(1, steps - steps_short)          # what a step costs once the call's fixed costs are paid
    # ... and ONE epoch over a file no call has seen (a copy under another name: no parsed copy in memory, no .dctr.npz beside it): the
    # text really is parsed inside this call -- what TextLineDataset.map(decode_libsvm) pays on every epoch of the reference
    import shutil
    cold = os.path.join(d, "cold.libsvm")
    shutil.copyfile(path, cold)
    t0 = time.perf_counter()
    est.train(input_fn=lambda: mod.input_fn([cold], num_epochs=1, batch_size=B))
    torch.cuda.synchronize()
    dt_cold = time.perf_counter() - t0
    # ... and the text really STREAMED: LibsvmDataset(streaming=True) (DCTR_INPUT_STREAMING=1) decodes the file in 64-MB chunks of whole
    # lines, chunk c + 1 on the library's thread team (num_parallel_calls = 10, DeepFM.py:84) while the batches of chunk c go through the
    # feeder's pinned buffers into the input slots and the steps run -- nothing cached, every epoch decodes the text again, like tf.data.
    # A file large enough that a pass lasts several hundred steps; steady state = the slope between a 1-epoch and a 3-epoch call.
    stream = None
    try:
        big_path = os.path.join(d, "stream.libsvm")
        reps = int(os.environ.get("DCTR_BENCH_STREAM_CHUNKS", "128"))            # x 32 768 lines (395 B each): 4.2 M lines, 1.6 GB of text
        with open(big_path, "w") as f:
            for _ in range(reps):
                f.write(chunk)
        n_big = 32768 * reps
        os.environ["DCTR_INPUT_STREAMING"] = "1"

        def timed_stream(ep):
            t0 = time.perf_counter()
            est.train(input_fn=lambda: mod.input_fn([big_path], num_epochs=ep, batch_size=B))
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        d1, d3 = timed_stream(1), timed_stream(3)
        per_epoch = (d3 - d1) / 2.0
        # the parser alone on the same file and thread count (what bounds the stream if it is the slower side)
        from tf_repos_amd.input_pipeline import parse_file
        t0 = time.perf_counter()
        parse_file(big_path, F, threads=10)
        d_parse = time.perf_counter() - t0
        stream = {"text_streaming_examples_per_sec": round(n_big / per_epoch, 1), "lines": n_big, "file_GB": round(os.path.getsize(big_path) / 1e9, 2),
                  "one_epoch_call_s": round(d1, 3), "three_epoch_call_s": round(d3, 3), "steady_ms_per_step": round(1e3 * per_epoch / (n_big // B), 4),
                  "one_epoch_call_examples_per_sec": round(n_big / d1, 1),
                  "parser_alone_lines_per_sec_10_threads": round(n_big / d_parse, 1),
                  "what": "Estimator.train over a %.1f-GB libsvm text file with DCTR_INPUT_STREAMING=1: decode (10 parser threads inside the library, chunk "
                          "by chunk) + batching + H2D + train steps overlapped, no cache; text_streaming_examples_per_sec = lines / ((3-epoch call - "
                          "1-epoch call) / 2); parser_alone = dctr_parse_libsvm_mt over the whole file with the same 10 threads, nothing else running"
                          % (os.path.getsize(big_path) / 1e9,)}
    except Exception as e:                                                     # noqa: BLE001
        stream = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    finally:
        os.environ.pop("DCTR_INPUT_STREAMING", None)
    shutil.rmtree(d, ignore_errors=True)
    return {"examples_per_sec": round(n_lines * epochs / dt, 1), "ms_per_step": round(1e3 * dt / steps, 4), "steps": steps, "wall_s": round(dt, 3),
            "text_streaming": stream,
            "steady_examples_per_sec": round(B / per_step, 1), "steady_ms_per_step": round(1e3 * per_step, 4),
            "fixed_cost_s": round(dt - per_step * steps, 3),
            "cold_text_one_epoch": {"examples_per_sec": round(n_lines / dt_cold, 1), "wall_s": round(dt_cold, 3), "steps": n_lines // B,
                                    "what": "one epoch of a libsvm file seen for the first time: the text IS parsed in the call (C parser, thread team), "
                                            "plus batching, H2D, the steps and the call's fixed costs (graph trace, lowering, 204 MB checkpoint)"},
            "what": "tf.estimator.Estimator.train (tf_shim) over a %d-line libsvm text file x %d epochs, batch %d.  The file's text is parsed ONCE, by "
                    "the short call before the timed one (which also writes the .dctr.npz binary cache beside it): the timed call loads the parsed "
                    "rows and replays its epochs from memory -- batching + H2D into the engine's input slots + train steps + checkpoint save (204 MB "
                    "of variables and Adam slots), NO text parse per epoch (cold_text_one_epoch has that).  examples_per_sec = wall time of the "
                    "whole call, steady_* = the slope between a %d-epoch and a %d-epoch call (the per-call fixed costs -- graph trace, lowering, "
                    "checkpoint -- cancel)" % (n_lines, epochs, B, short, epochs)}


def main():
    # stdout carries exactly ONE line, the JSON: everything else that native libraries print there (RCCL writes a version banner
    # to stdout at exit) is sent to stderr by pointing fd 1 at fd 2 and keeping the real stdout aside for the final line
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--table-mode", default="dense_exact", choices=["dense_exact", "touched_rows"])
    ap.add_argument("--driver", default=os.environ.get("DCTR_SHARD_DRIVER", "native"), choices=["native", "python"],
                    help="multi-GPU step driver: the C++ one over RCCL (default) or the torch.distributed orchestration; a native driver that cannot start falls back LOUDLY (stderr + config.driver), DCTR_BENCH_STRICT=1 forbids it")
    ap.add_argument("--selftest", action="store_true", help="multi-GPU: first check that the N-rank loss of step 0 equals one rank's on the same global batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-classic-reference", action="store_true", help="skip the 200-step reference run with the classic table sweep")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the Estimator.train block (text parse + H2D + steps)")
    ap.add_argument("--cpu-steps", type=int, default=200)
    ap.add_argument("--uniform-ids", action="store_true", help="uniform ids instead of Zipf (cache-worst case)")
    ap.add_argument("--feature-size", type=int, default=0, help="override the config's vocabulary (tools/c5_shard_projection.py: ONE shard of c5's table, V / 8 rows, on one GPU)")
    ap.add_argument("--gemm-mode", default=os.environ.get("DCTR_BENCH_GEMM_MODE", "split"), choices=["split", "exact"],
                    help="arithmetic of the MLP products (dctr_config.gemm_mode): split = three bf16 planes, six products, f32 accumulate (f32-equivalent, "
                         "the whole GPU suite passes in it at the exact mode's tolerances); exact = f32-input MFMA.  The line reports the other mode's step time too")
    ap.add_argument("--sweep-period", type=int, default=0, help="dense_exact + Adam: period of the time-blocked table sweep (0 = library default, 1 = classic: every row every step)")
    args = ap.parse_args()

    import numpy as np
    import torch
    from tf_repos_amd.engine import Engine, EngineConfig
    from tf_repos_amd.synth import synth_batch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    w = dict(CONFIGS[args.config])
    if args.feature_size > 0:
        w["feature_size"] = args.feature_size
        w["name"] += " [feature_size overridden: %d]" % args.feature_size
    w["table_sweep_period"] = args.sweep_period          # (row-sharded runs: the native driver's owner side, csrc/lag.h)
    # (--selftest keeps the workload's dropout: a mask is a function of the GLOBAL example row -- StepState::row0 -- so N ranks draw
    #  exactly what one rank draws on the same global batch)
    B, F, K, V = w["batch"], w["field_size"], w["embedding_size"], w["feature_size"]
    big = V * (K + 1) * 4 > (2 << 30)           # tables that must be initialised on the device

    sharded = world > 1 or bool(os.environ.get("DCTR_FORCE_SHARDED"))     # the env var exercises the RCCL path on one GPU
    os.environ["DCTR_GEMM_MODE"] = args.gemm_mode        # (every engine of this process, the row-sharded trainer's included)
    if sharded:
        import torch.distributed as dist
        from tf_repos_amd.distributed import ShardedTrainer
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        # (no silent fallback: a native driver that cannot start is an error, and the JSON line says which driver ran)
        driver_note = None
        try:
            trainer = ShardedTrainer(w, rank, world, dev, table_mode=args.table_mode, driver=args.driver, init_tables=not big)
        except Exception as e:                                                                # noqa: BLE001
            # NOT silent: the failure goes to stderr, the JSON line names the driver that actually ran and why; DCTR_BENCH_STRICT=1
            # turns it into an error.  (The Python orchestration runs the same protocol with the same results, slower on the host.)
            if args.driver != "native" or os.environ.get("DCTR_BENCH_STRICT") == "1":
                raise
            print("rank %d: NATIVE SHARDED DRIVER FAILED (%s: %s) -- falling back to the torch.distributed orchestration" % (rank, type(e).__name__, e),
                  file=sys.stderr, flush=True)
            driver_note = "python (torch.distributed orchestration; the native driver failed to start: %s)" % (str(e)[:200],)
            args.driver = "python"
            trainer = ShardedTrainer(w, rank, world, dev, table_mode=args.table_mode, driver="python", init_tables=not big)
        if big:                                  # c5: tables drawn on the device, shard by shard
            for name in ("emb", "linear"):
                fill_normal_(trainer.eng.param_tensor(name), 0.01, 1000 + rank)
        eng = trainer.eng
        step = lambda i, v, l, nxt: trainer.train_step(i, v, l, next_ids=nxt)      # routes the next batch's ids a step ahead
        barrier = dist.barrier
    else:
        eng = Engine(EngineConfig(model=w["model"], field_size=F, feature_size=V, embedding_size=K, deep_layers=w["deep_layers"],
                                  dropout=w["dropout"], l2_reg=w["l2_reg"], learning_rate=w["learning_rate"], optimizer=w["optimizer"],
                                  table_mode=args.table_mode, max_batch=B, seed=1, table_sweep_period=args.sweep_period,
                                  gemm_mode=args.gemm_mode, use_graph=os.environ.get("DCTR_USE_GRAPH", "0") == "1"))
        rng = np.random.default_rng(1)
        for name, shp in eng.param_shapes.items():
            if big and name in ("emb", "linear"):
                fill_normal_(eng.param_tensor(name), 0.01, 1000)
            else:
                eng.set_param(name, rng.normal(0, 0.01, size=shp).astype(np.float32))
        if os.environ.get("DCTR_BENCH_PREFETCH", "1") == "1":      # the input pipeline's hint: next batch's ids grouped a step ahead
            def step(i, v, l, nxt):
                eng.train_step(i, v, l, want_loss=False)
                eng.prefetch_ids(nxt)
        else:
            step = lambda i, v, l, nxt: eng.train_step(i, v, l, want_loss=False)
        barrier = lambda: None

    # watchdog: a wedged collective (multi-GPU runs are launched by the driver, not from here) must end the job with a message,
    # not sit until an outer timeout
    import threading
    done = threading.Event()

    def watchdog(limit=float(os.environ.get("DCTR_BENCH_TIMEOUT", "900"))):
        if not done.wait(limit):
            import faulthandler
            print("rank %d: bench did not finish within %.0f s -- aborting (stacks follow)" % (rank, limit), file=sys.stderr, flush=True)
            faulthandler.dump_traceback(file=sys.stderr)
            os._exit(3)
    threading.Thread(target=watchdog, daemon=True).start()

    nb = 8
    batches = []
    host_batches = []
    for i in range(nb):
        ids, vals, labels = synth_batch(B, F, V, seed=20260924 + 1 + rank * 1000 + i, uniform_ids=args.uniform_ids)
        if i == 0:
            host_batches.append((ids, vals, labels))
        t = (torch.from_numpy(ids).to(dev), torch.from_numpy(vals).to(dev), torch.from_numpy(labels).to(dev))
        # resident inputs: the 8 synthetic batches live in the engine's 8 input slots (what the input pipeline's H2D copy
        # targets), so a step reads them in place
        si, sv, sl = eng.input_slot(i)
        si[:B].copy_(t[0]); sv[:B].copy_(t[1]); sl[:B].copy_(t[2])
        batches.append((si[:B], sv[:B], sl[:B]))

    if args.selftest:
        # N ranks on the global batch of step 0 == ONE rank on the same global batch, over the real transport (the workload's own keep_prob; both
        # sides draw the weights from seed 1).  A check, not a measurement: prints its own JSON line and exits.
        if not sharded:
            raise SystemExit("--selftest needs --gpus N > 1 (or DCTR_FORCE_SHARDED=1)")
        # (c5: rank 0 also holds the WHOLE 1e8-row table of the one-rank reference engine -- 12.9 GB of parameters + 25.8 GB of Adam slots
        #  beside its own shard: 288 GB of HBM take it)
        import torch.distributed as dist
        gathered = [None] * world
        dist.all_gather_object(gathered, host_batches[0])
        loss_n = trainer.train_step(*batches[0], want_loss=True, next_ids=batches[1][0])
        ok, loss_1 = True, None
        if rank == 0:
            gi, gv, gl = (np.concatenate([g[k] for g in gathered]) for k in range(3))
            ref = Engine(EngineConfig(model=w["model"], field_size=F, feature_size=V, embedding_size=K, deep_layers=w["deep_layers"],
                                      dropout=w["dropout"], l2_reg=w["l2_reg"], learning_rate=w["learning_rate"],
                                      optimizer=w["optimizer"], table_mode=args.table_mode, max_batch=B * world, seed=1))
            r1 = np.random.default_rng(1)
            for name, shp in ref.param_shapes.items():
                if big and name in ("emb", "linear"):
                    # the shards' own draws (fill_normal_ with seed 1000 + r on a contiguous [rows of shard r, ...] tensor), laid into rows r, r + N, ...
                    full = ref.param_tensor(name)
                    for r_ in range(world):
                        n_r = (V - r_ + world - 1) // world
                        tmp = torch.empty((n_r,) + tuple(full.shape[1:]), device=dev)
                        fill_normal_(tmp, 0.01, 1000 + r_)
                        full[r_::world].copy_(tmp)
                        del tmp
                    continue
                ref.set_param(name, r1.normal(0, 0.01, size=shp).astype(np.float32))
            loss_1 = ref.train_step(torch.from_numpy(gi).to(dev), torch.from_numpy(gv).to(dev), torch.from_numpy(gl).to(dev))
            ref.close()
            ok = abs(loss_n - loss_1) <= 1e-5 * max(1.0, abs(loss_1))
            os.write(real_stdout, (json.dumps({"selftest": "ok" if ok else "MISMATCH", "n_gpus": world, "driver": args.driver,
                                               "loss_n_ranks": loss_n, "loss_one_rank": loss_1, "global_batch": B * world}) + "\n").encode())
        done.set()
        trainer.close()
        dist.destroy_process_group()
        sys.exit(0 if ok else 1)

    # (the in-step timer's event pool is created by its first enable: here, not between the warm-up and the timed region, where the
    #  milliseconds it takes would leave the GPU idle long enough to drop its clocks)
    eng.step_timer(1 if os.environ.get("DCTR_BENCH_TIMER") == "1" else 2)
    # this box's HBM roofline (1 GiB float4 copy, read + write; reported as hbm_measured_copy_GBps) is measured HERE, right before the
    # warm-up steps, not after the timed region: building the engine and loading its 1e6-row tables leaves the GPU idle for seconds, and
    # a GPU that has idled for >= 50 ms runs its next few milliseconds below its working clocks (tools/fixed_cost_probe.py: +0.33 ms on a
    # 20-step block) -- the driver's invocation is 5 + 20 steps, 6 ms in all.  7 ms of copy kernels put the clocks where any run longer
    # than that finds them; nothing inside the timed region changes.
    copy_gbps = eng.measure_copy_bandwidth(1 << 30, 20)
    for s in range(args.warmup):
        step(*batches[s % nb], batches[(s + 1) % nb][0])
    barrier()
    torch.cuda.synchronize()
    # hipEvents on the roofline kernel inside the timed steps: every forward layer's launch carries its own start / stop events
    # (DCTR_BENCH_TIMER=1: rounds 1-3's bracket of two records around layer 0 -- two barrier packets inside the interval)
    timer_mode = 1 if os.environ.get("DCTR_BENCH_TIMER") == "1" else 2
    if os.environ.get("DCTR_BENCH_TIMER") == "0":          # (A/B: what the in-step timer's sampled steps cost the timed region)
        eng.step_timer(False)
    else:
        eng.step_timer(timer_mode)
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(*batches[(args.warmup + s) % nb], batches[(args.warmup + s + 1) % nb][0])
    eng.sync_tables()                         # time-blocked table sweep: every row's updates of the timed steps are computed INSIDE the timed region
    t_enq = time.perf_counter() - t0          # host time to ENQUEUE the steps (the GPU may still be running)
    barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    in_ms, in_n = eng.step_timer(False)
    nl = len(w["deep_layers"])
    in_layers = [eng.step_timer_layer(i) for i in range(3 * nl)] if timer_mode == 2 else []      # forward, dgrad, wgrad of every layer
    # the run's fixed part: a second, longer block of the same loop gives the slope (steady state) and, by difference, what the
    # closing flush of the lagging rows costs -- the driver's 20-step number and a 200-step number then agree by construction
    steady_ms, flush_ms = None, None
    if not sharded:
        n2 = max(5 * args.steps, 200)
        t1 = time.perf_counter()
        for s in range(n2):
            step(*batches[(args.warmup + s) % nb], batches[(args.warmup + s + 1) % nb][0])
        eng.sync_tables()
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t1
        steady_ms = 1e3 * (el2 - el) / (n2 - args.steps)
        flush_ms = 1e3 * el - args.steps * steady_ms
    if sharded:
        import torch.distributed as dist
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    out = None
    if rank == 0:
        out = {
            "metric": "examples/sec DeepFM Criteo-39-field batch %d" % B, "value": round(B * world * args.steps / el, 1),
            "unit": "examples/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * el / args.steps, 4), "host_enqueue_ms_per_step": round(1e3 * t_enq / args.steps, 4),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE[args.gemm_mode], "data": "synthetic Criteo-shaped (Zipf categorical ids), random-init weights",
            "config": {"workload": w["name"], "config": args.config, "global_batch": B * world, "table_mode": args.table_mode,
                       "parallelism": "single GPU" if world == 1 else "row-sharded tables (id %% %d) + data-parallel dense" % world,
                       "driver": ("single-GPU engine" if not sharded else (driver_note or (args.driver + (" (C++ step driver over RCCL)" if args.driver == "native" else " (torch.distributed orchestration)")))),
                       "ids": "uniform" if args.uniform_ids else "zipf", "gemm_mode": args.gemm_mode,
                       "next_batch_hint": bool(not sharded and os.environ.get("DCTR_BENCH_PREFETCH", "1") == "1"),
                       "table_sweep_period": (args.sweep_period or int(os.environ.get("DCTR_SWEEP_PERIOD", "8"))) if (args.table_mode == "dense_exact" and (not sharded or args.driver == "native")) else 1},
        }
        if steady_ms is not None:
            # ms_per_step = (steps x steady_ms_per_step + final_flush_ms) / steps: the timed region closes with the flush that brings every
            # lagging table row to the present (its deferred updates are paid inside the region)
            out["steady_ms_per_step"] = round(steady_ms, 4)
            out["final_flush_ms"] = round(flush_ms, 4)
            out["steady_examples_per_sec"] = round(B * world / (steady_ms * 1e-3), 1)
        # ---- per-stage timing (hipEvents around a graph of back-to-back launches, on torch's current stream)
        e = eng
        rows = (V + world - 1) // world                            # rows of this rank's table shard
        names = ["opt_table", "mlp0_fwd", "mlp0_dgrad", "mlp0_wgrad"]
        if not sharded:
            names = ["embed_gather", "forward", "head", "backward_dense", "group_ids", "scatter", "tail", "opt_dense"] + names
        stages = {name: e.time_stage(name, iters=(30 if not big else 3)) for name in names}
        gather_bytes = B * (F * (12 + 8 * K) + 8)                 # algorithmic bytes of the gather
        # dense-exact table step as implemented: theta,m,v read + write (6 streams) + the 4-byte slot word per row; the per-row
        # gradient is NOT a dense stream here (only the ~U touched rows read a compact gradient row), so SURVEY 8d's 7-stream
        # figure (7*V*(K+1)*4) would flatter the kernel -- 6 streams is what the algorithm must move
        table_bytes = 6 * rows * (K + 1) * 4 + 4 * rows
        mlp0_flops = 2.0 * B * (F * K) * w["deep_layers"][0]
        kernels = {
            # (the CLASSIC sweep -- every row of the table through HBM -- timed alone; the timed steps run the time-blocked sweep instead:
            #  roofline.hbm_kernel below)
            "opt_table_dense_adam_classic": {"bound": "hbm", "ms": stages["opt_table"], "achieved": table_bytes / stages["opt_table"] / 1e6,
                                             "peak": HBM_PEAK_GBS, "unit": "GB/s"},
            "mlp0_fwd_gemm": {"bound": "mfma", "ms": stages["mlp0_fwd"], "achieved": mlp0_flops / stages["mlp0_fwd"] / 1e9,
                              "peak": MFMA_F32_PEAK_TFS, "unit": "TFLOP/s"},
            "mlp0_dgrad_gemm": {"bound": "mfma", "ms": stages["mlp0_dgrad"], "achieved": mlp0_flops / stages["mlp0_dgrad"] / 1e9,
                                "peak": MFMA_F32_PEAK_TFS, "unit": "TFLOP/s"},
            "mlp0_wgrad_gemm": {"bound": "mfma", "ms": stages["mlp0_wgrad"], "achieved": mlp0_flops / stages["mlp0_wgrad"] / 1e9,
                                "peak": MFMA_F32_PEAK_TFS, "unit": "TFLOP/s"},
        }
        if not sharded:
            kernels["embed_gather_fwd_step_table"] = {"bound": "hbm", "ms": stages["embed_gather"], "achieved": gather_bytes / stages["embed_gather"] / 1e6,
                                                      "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                      "note": "the step's own table (%d MB%s)" % (V * (K + 1) * 4 >> 20, ", resident in the 256 MB Infinity Cache" if V * (K + 1) * 4 < (256 << 20) else "")}
            if not big:
                g_ms, g_bytes = hbm_resident_gather(dev, K=K, B=B, F=F)
                kernels["embed_gather_fwd"] = {"bound": "hbm", "ms": g_ms, "achieved": g_bytes / g_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                               "note": "HBM-resident: 64 M-row table (4.3 GB), uniform ids, the op through the C ABI, back-to-back launches"}
                g_ms, g_bytes = hbm_resident_gather(dev, K=32, V=32 * 1024 * 1024, B=B, F=F)
                kernels["embed_gather_fwd_k32_hbm"] = {"bound": "hbm", "ms": g_ms, "achieved": g_bytes / g_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                       "note": "c5's row shape: K = 32 (a row = one 128-byte granule), 32 M-row table (4.4 GB), uniform ids; layouts compared in profiles/r03_gather_layouts.txt"}
                # the distribution the path actually sees at c5 (BASELINE configs[4]): ONE shard of the 1e8-row table on a GPU (1.25e7 rows,
                # K = 32: 1.6 GB + Adam slots elsewhere), 8192 examples per step -- Zipf ids from the workload's generator, and uniform ids
                # over the same shard as the worst case (round-5 verdict item 6)
                for tag, zf in (("zipf", True), ("uniform", False)):
                    g_ms, g_bytes = hbm_resident_gather(dev, K=32, V=12_500_000, B=8192, F=F, iters=100, zipf=zf)
                    kernels["embed_gather_fwd_c5_shard_%s" % tag] = {
                        "bound": "hbm", "ms": g_ms, "achieved": g_bytes / g_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "note": "c5's shard shape: 1.25e7 rows x K = 32 (1.6 GB table), B = 8192, %s ids" % ("Criteo-shaped Zipf (synth_batch)" if zf else "uniform")}
        out["hbm_measured_copy_GBps"] = round(copy_gbps, 1)        # (measured before the warm-up steps, see there)
        # the gathers, memory side: a random access costs a 128-byte granule whatever it asks for (profiles/r02_gather_hbm_pmc.txt: 41.2 MB
        # fetched per launch at K = 16 AND at K = 32) -- one per row piece of <= 128 B and one per 4-byte linear weight -- plus e written
        for kn, kk, bb in (("embed_gather_fwd", K, B), ("embed_gather_fwd_k32_hbm", 32, B), ("embed_gather_fwd_c5_shard_zipf", 32, 8192),
                           ("embed_gather_fwd_c5_shard_uniform", 32, 8192)):
            if kn in kernels:
                # (an upper bound for Zipf ids: repeated hot rows are served by the caches, not by a granule each)
                ms_side = bb * F * (128 * ((4 * kk + 127) // 128) + 128) + bb * F * 4 * kk
                kernels[kn]["memory_side_bytes"] = int(ms_side)
                kernels[kn]["memory_side_GBps"] = round(ms_side / kernels[kn]["ms"] / 1e6, 1)
                kernels[kn]["frac_memory_side_of_measured_copy"] = round(ms_side / kernels[kn]["ms"] / 1e6 / copy_gbps, 4)
        for k in kernels.values():
            if k["bound"] == "hbm":
                k["frac_of_measured_copy"] = round(k["achieved"] / copy_gbps, 4)
            k["frac"] = round(k["achieved"] / k["peak"], 4)
            k["achieved"] = round(k["achieved"], 2)
            k["ms"] = round(k["ms"], 5)
        # The dominant kernel family of the step is the MLP GEMM: nine products per step (forward, dgrad, weight gradient of three
        # layers).  `roofline` is the kernel TEMPLATE with the most time in the timed steps -- every product's launch carries its own
        # start / stop events (dctr_step_timer mode 2, every 32nd step) -- and `roofline.family` all nine; `kernels` holds layer 0's
        # products timed alone, back to back.
        split = args.gemm_mode == "split"
        peak = MFMA_SPLIT_PEAK_TFS if split else MFMA_F32_PEAK_TFS
        for kn in ("mlp0_fwd_gemm", "mlp0_dgrad_gemm", "mlp0_wgrad_gemm"):
            kernels[kn]["peak"] = round(peak, 1)
            kernels[kn]["frac"] = round(kernels[kn]["achieved"] / peak, 4)
            kernels[kn]["frac_of_f32_mfma_peak"] = round(kernels[kn]["achieved"] / MFMA_F32_PEAK_TFS, 4)
        dims = [F * K] + list(w["deep_layers"])
        layer_flops = [2.0 * B * dims[i] * dims[i + 1] for i in range(nl)]

        def template(kind, i):          # (the names c2's shapes take; other configs: labels only)
            if split:
                # (forward and the 400-wide dgrads: 2 x 7 tiles at two blocks per CU unless DCTR_DR3_SMALL says otherwise -- csrc/gemm_dr3.hip)
                sm = os.environ.get("DCTR_DR3_SMALL", "fd")
                f_t, d_t, w_t = ("2, 7" if c_ in sm and sm != "none" else "4, 7" for c_ in "fdw")
                # (template arguments: TM, TN, A_RC, B_RC, CS, EPI, B_PRE, A_PRE -- csrc/gemm_dr.h)
                return {"fwd": "void dctr::gemm_dr3_kernel<%s, true, true, false, 1, true, false>" % f_t,
                        "dgrad": ("void dctr::gemm_dr3_kernel<%s, true, true, false, 2, true, false>" % d_t) if i > 0 else "void dctr::gemm_dr3_kernel<4, 10, true, true, false, 0, true, false>",
                        "wgrad": "void dctr::gemm_dr3_kernel<%s, false, false, true, 0, false, false>" % w_t}[kind]
            return {"fwd": "void dctr::gemm_dr_kernel<2, 13, true, false, false, 1, 0>",
                    "dgrad": "void dctr::gemm_dr_kernel<2, 13, true, true, false, 2, 0>" if i > 0 else "void dctr::gemm_dr_kernel<4, 10, true, true, false, 0, 0>",
                    "wgrad": "void dctr::gemm_dr_kernel<2, 13, false, false, true, 0, 0>"}[kind]

        def product_bytes(kind, i):     # what one launch must move: both operands once, the result once (split: the weight as 3 bf16 planes)
            wb = 6 if split else 4
            if kind == "wgrad":
                return 4 * B * dims[i] + 4 * B * dims[i + 1] + 4 * dims[i] * dims[i + 1]
            return 4 * B * dims[i] + wb * dims[i] * dims[i + 1] + 4 * B * dims[i + 1]
        products = []
        if in_n > 0 and timer_mode == 2 and len(in_layers) == 3 * nl:
            for ki, kind in enumerate(("fwd", "dgrad", "wgrad")):
                for i in range(nl):
                    ms_i, n_i = in_layers[ki * nl + i]
                    if n_i > 0:
                        products.append({"product": "%s layer %d" % (kind, i), "shape": "%dx%dx%d" % (B, dims[i], dims[i + 1]), "template": template(kind, i).replace("void dctr::", ""),
                                         "ms": round(ms_i, 5), "launches": n_i, "flops": layer_flops[i], "bytes": product_bytes(kind, i)})
        r = dict(kernels["mlp0_fwd_gemm"])
        gemm_name = template("fwd", 1)
        r["ms_alone_layer0_fwd"] = r.pop("ms")
        if products:
            groups = {}
            for pr in products:
                groups.setdefault(pr["template"], []).append(pr)
            # (per launch-weighted: sum_i n_i flops_i / sum_i n_i ms_i -- the launch counts of the products may differ when the event pool fills)
            def agg(prs):
                nn = sum(p_["launches"] for p_ in prs)
                tf = sum(p_["launches"] * p_["flops"] for p_ in prs) / sum(p_["launches"] * p_["ms"] for p_ in prs) / 1e9
                return nn, tf, sum(p_["ms"] for p_ in prs), sum(p_["launches"] * p_["ms"] for p_ in prs) / nn
            dom_t = max(groups, key=lambda t: sum(p_["ms"] for p_ in groups[t]))
            nn, tf, ms_step, ms_mean = agg(groups[dom_t])
            gemm_name = "void dctr::" + dom_t
            r.update({"kernel": "%s -- %s: the template with the most time in the timed steps (%.1f us of the nine products' %.1f us per step)" % (
                          dom_t, ", ".join(p_["product"] for p_ in groups[dom_t]), 1e3 * ms_step, 1e3 * sum(p_["ms"] for p_ in products)),
                      "ms": round(ms_mean, 5), "achieved": round(tf, 2), "peak": round(peak, 1), "frac": round(tf / peak, 4),
                      "frac_of_f32_mfma_peak": round(tf / MFMA_F32_PEAK_TFS, 4), "launches_timed": nn,
                      "method": "hipExtLaunchKernel start/stop events on each timed dispatch (every 32nd step), launch-weighted flops / duration over the template's launches",
                      "algorithmic_bytes": int(sum(p_["bytes"] for p_ in groups[dom_t]) / len(groups[dom_t]))})
            an, atf, ams, _ = agg(products)
            r["family"] = {"what": "all nine MLP products of the step (%s)" % ("gemm_dr3_kernel: split precision" if split else "gemm_dr_kernel: exact f32 MFMA"),
                           "achieved": round(atf, 2), "peak": round(peak, 1), "frac": round(atf / peak, 4), "frac_of_f32_mfma_peak": round(atf / MFMA_F32_PEAK_TFS, 4),
                           "us_per_step_sum_of_dispatches": round(1e3 * ams, 2), "launches_timed": an,
                           "by_template": {t: {"products": [p_["product"] for p_ in g], "us_per_step": round(1e3 * sum(p_["ms"] for p_ in g), 2),
                                               "achieved": round(agg(g)[1], 2), "frac": round(agg(g)[1] / peak, 4)} for t, g in groups.items()}}
            r["layers"] = [{k_: v_ for k_, v_ in pr.items() if k_ not in ("flops", "bytes")} | {"frac": round(pr["flops"] / pr["ms"] / 1e9 / peak, 4)} for pr in products]
            if split:
                r["peak_note"] = ("f32-equivalent flops against the dense bf16 MFMA peak / 6 (%.0f / 6 = %.1f TF): one f32 multiply-add is six bf16 plane "
                                  "products; frac_of_f32_mfma_peak is the same rate against the f32-input MFMA's %.1f TF" % (MFMA_BF16_PEAK_TFS, peak, MFMA_F32_PEAK_TFS))
        elif in_n > 0:
            r["kernel"] = "mlp0_fwd_gemm (layer 0: %dx%dx%d)" % (B, F * K, w["deep_layers"][0])
            r["ms"] = round(in_ms, 5)
            r["achieved"] = round(mlp0_flops / in_ms / 1e9, 2)
            r["frac"] = round(r["achieved"] / peak, 4)
            r["launches_timed"] = in_n
            r["method"] = "two hipEvent records around layer 0's launch (a bracket: holds two barrier packets)"
            r["algorithmic_bytes"] = product_bytes("fwd", 0)
        else:
            r["kernel"] = "mlp0_fwd_gemm alone (no in-step timer on this path)"
            r["ms"] = r["ms_alone_layer0_fwd"]
            r["algorithmic_bytes"] = product_bytes("fwd", 0)
        # HBM bytes per launch of exactly this kernel template (PMC passes of tools/profile_round.sh, committed under profiles/)
        r["traffic"] = pmc_traffic_bytes(gemm_name) if (not sharded and args.config == "c2") else None
        fam = rocprof_family("void dctr::gemm_dr3_kernel" if split else "void dctr::gemm_dr_kernel") if (not sharded and args.config == "c2") else {}
        if fam and products:
            # the same figures by rocprofv3's kernel timestamps of this command (its launches include the ~100 alone-timed ones of layer 0)
            mean_flops = {}
            for pr in products:
                mean_flops.setdefault("void dctr::" + pr["template"], []).append(pr["flops"])
            tot_us = sum(us_ for nm, (us_, _c) in fam.items() if nm in mean_flops)
            tot_fl = sum(c_ * sum(mean_flops[nm]) / len(mean_flops[nm]) for nm, (_u, c_) in fam.items() if nm in mean_flops)
            if tot_us > 0:
                r["family"]["frac_rocprof"] = round(tot_fl / (tot_us * 1e-6) / 1e12 / peak, 4)
            if gemm_name in fam and fam[gemm_name][1] > 0:
                us = fam[gemm_name][0] / fam[gemm_name][1]
                r["rocprof_avg_us"] = round(us, 2)
                r["frac_rocprof"] = round(sum(mean_flops[gemm_name]) / len(mean_flops[gemm_name]) / (us * 1e-6) / 1e12 / peak, 4)
        stale = [f for f in (PMC_FILE, STATS_FILE) if profile_is_stale(f)]
        if stale:
            r["profile_warning"] = "older than the built library, re-run tools/profile_round.sh: " + ", ".join(stale)
        # the step's largest HBM stream AS THE TIMED STEPS RUN IT: the background sweep over 1/N of the table per step (csrc/lag.h),
        # by rocprofv3's in-step duration and PMC bytes of exactly that kernel (committed summaries of this same command)
        period = out["config"]["table_sweep_period"]
        sweep_name = "void dctr::(anonymous namespace)::lag_advance_kernel<%d, false, 4, 1>" % (K // 4)
        if period > 1 and not sharded:
            sweep_bytes = (6 * (K + 1) * 4 + 4 + 1) * ((rows + period - 1) // period)      # theta, m, v read + write of 1/N of the rows, slot word, stamp
            sweep_us = rocprof_avg_us(sweep_name) if args.config == "c2" else None
            hk = {"kernel": "lag_advance_kernel<%d, false, 4, 1> (background sweep of 1/%d of the table per step, lagging rows replayed in registers)" % (K // 4, period),
                  "bound": "hbm", "algorithmic_bytes": int(sweep_bytes), "traffic": pmc_traffic_bytes(sweep_name) if args.config == "c2" else None,
                  "us_in_step": sweep_us, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "achieved": round(sweep_bytes / sweep_us / 1e3, 2) if sweep_us else None,
                  "frac": round(sweep_bytes / sweep_us / 1e3 / HBM_PEAK_GBS, 4) if sweep_us else None,
                  "element_updates_per_launch": int(rows * (K + 1)),
                  # (a VALU op takes a wave 4 cycles on a SIMD, sqrt / rcp 16: ~9.5 + 2 per element-update with the identity-coefficient replay loop)
                  "alu_floor_us": round(rows * (K + 1) * (9.5 * 4 + 2 * 16) / 64.0 / 1024 / 2.1e3, 2),
                  "note": "NOT an HBM-bound kernel (frac is its traffic against HBM for reference only): every swept row replays %d Adam steps in "
                          "registers -- V (K+1) element-updates per launch however scheduled, ~9.5 VALU issue slots of 4 cycles + sqrt and rcp at 16 per "
                          "element-update and wave = alu_floor_us on the chip's 1024 SIMDs (PMC, profiles/r06_lag_pmc.txt: SQ_ACTIVE_INST_VALU 4.2 M quad-"
                          "cycles per launch = 7.8 us); the rest is memory latency the replay does not cover (SQ_WAIT_ANY 44 %% of the wave cycles).  "
                          "Round 6 measured a software-pipelined kernel, a <= 64-register one-row-per-lane kernel and 1 / 4 / 8 blocks per CU, alone "
                          "(profiles/r06_lag_probe.txt: 18-22 us whatever the shape) and in the step (r06_ab_schedule_and_sweep_knobs.txt): none faster.  The "
                          "classic sweep it replaces: kernels.opt_table_dense_adam_classic" % period}
        else:
            hk = dict(kernels["opt_table_dense_adam_classic"])
            hk["traffic"] = pmc_traffic_bytes("void dctr::opt_table_kernel<0, 4, true>") if (not sharded and args.config == "c2") else None
        r["hbm_kernel"] = hk
        # the whole step against the matrix pipes: forward + dgrad + wgrad flops of the MLP over ms_per_step
        step_flops = 3 * sum(2.0 * B * dims[i] * dims[i + 1] for i in range(len(dims) - 1))
        r["step_mfma_frac"] = round(step_flops / (out["ms_per_step"] * 1e-3) / 1e12 / peak, 4)
        r["step_mfma_frac_of_f32_peak"] = round(step_flops / (out["ms_per_step"] * 1e-3) / 1e12 / MFMA_F32_PEAK_TFS, 4)
        r["step_gemm_gflop"] = round(step_flops / 1e9, 3)
        r["traffic_source"] = PMC_FILE + ": rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch of the exactly named kernel (gfx950 FETCH_SIZE counts 1/2 of 16-B/lane streams; MI355X_MICROARCH.md HBM section)"
        out["roofline"] = r
        out["kernels"] = kernels
        out["stage_ms"] = {k: round(v, 5) for k, v in stages.items()}
        if not sharded and not big and out["config"]["table_sweep_period"] > 1 and not args.no_classic_reference:
            # the same workload with the classic sweep (every table row through HBM every step), for reference beside `value`
            ref = Engine(EngineConfig(model=w["model"], field_size=F, feature_size=V, embedding_size=K, deep_layers=w["deep_layers"],
                                      dropout=w["dropout"], l2_reg=w["l2_reg"], learning_rate=w["learning_rate"], optimizer=w["optimizer"],
                                      table_mode=args.table_mode, max_batch=B, seed=1, table_sweep_period=1))
            r1 = np.random.default_rng(1)
            for name, shp in ref.param_shapes.items():
                ref.set_param(name, r1.normal(0, 0.01, size=shp).astype(np.float32))
            rb = []
            for i in range(nb):
                si, sv, sl = ref.input_slot(i)
                si[:B].copy_(batches[i][0]); sv[:B].copy_(batches[i][1]); sl[:B].copy_(batches[i][2])
                rb.append((si[:B], sv[:B], sl[:B]))
            for s_ in range(20 + 200):
                if s_ == 20:
                    torch.cuda.synchronize()
                    tr0 = time.perf_counter()
                ref.train_step(*rb[s_ % nb], want_loss=False)
                ref.prefetch_ids(rb[(s_ + 1) % nb][0])
            torch.cuda.synchronize()
            out["classic_sweep_ms_per_step"] = round(1e3 * (time.perf_counter() - tr0) / 200, 4)
            ref.close()
        if not sharded and not big and not args.no_classic_reference:
            # the same loop in the OTHER arithmetic of the MLP products (dctr_config.gemm_mode), for reference beside `value`
            other = "exact" if args.gemm_mode == "split" else "split"
            os.environ["DCTR_GEMM_MODE"] = other
            ref = Engine(EngineConfig(model=w["model"], field_size=F, feature_size=V, embedding_size=K, deep_layers=w["deep_layers"],
                                      dropout=w["dropout"], l2_reg=w["l2_reg"], learning_rate=w["learning_rate"], optimizer=w["optimizer"],
                                      table_mode=args.table_mode, max_batch=B, seed=1, table_sweep_period=args.sweep_period, gemm_mode=other))
            os.environ["DCTR_GEMM_MODE"] = args.gemm_mode
            r1 = np.random.default_rng(1)
            for name, shp in ref.param_shapes.items():
                ref.set_param(name, r1.normal(0, 0.01, size=shp).astype(np.float32))
            rb = []
            for i in range(nb):
                si, sv, sl = ref.input_slot(i)
                si[:B].copy_(batches[i][0]); sv[:B].copy_(batches[i][1]); sl[:B].copy_(batches[i][2])
                rb.append((si[:B], sv[:B], sl[:B]))
            for s_ in range(20 + 200):
                if s_ == 20:
                    torch.cuda.synchronize()
                    tr0 = time.perf_counter()
                ref.train_step(*rb[s_ % nb], want_loss=False)
                ref.prefetch_ids(rb[(s_ + 1) % nb][0])
            ref.sync_tables()
            torch.cuda.synchronize()
            out["gemm_mode_%s_ms_per_step" % other] = round(1e3 * (time.perf_counter() - tr0) / 200, 4)
            out["gemm_mode_%s_dtype" % other] = DTYPE[other]
            ref.close()
        if not sharded and not big and not args.no_end_to_end:
            try:
                out["end_to_end"] = end_to_end(w)
            except Exception as e:                                                     # noqa: BLE001  (a reported extra, never `value`)
                out["end_to_end"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        if not sharded and not args.no_cpu_baseline and not big:
            out["cpu_baseline"] = cpu_baseline(w, steps=args.cpu_steps)
            out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
            out["speedup_vs_cpu_compute_only"] = round(out["value"] / out["cpu_baseline"]["compute_only"], 1)
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    done.set()
    if sharded:
        import torch.distributed as dist
        trainer.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
