#!/usr/bin/env python
"""bench.py -- examples/sec of the DeepFM training step (BASELINE.json metric) on N MI355X GPUs of one node.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1], SURVEY 8d "c2"): DeepFM, 39 fields, vocab 1e6, emb_dim 16, batch 4096 per GPU
(weak scaling), MLP 400-400-400 with keep_prob 0.5, Adam lr 5e-4, l2_reg 1e-4 (deep_ctr/README.md:49), f32 arithmetic,
synthetic Criteo-shaped libsvm-equivalent tensors already resident in HBM, random-init N(0,0.01) weights.
A "step" = forward + loss + backward + optimizer over one batch, table optimizer in DENSE-EXACT mode (what the
reference's TF graph does: l2_loss on the tables makes the optimizer stream all V rows every step).
Prints ONE JSON line (rank 0).  The `roofline` object is for the kernel that dominates the step; `cpu_baseline` is the
torch-CPU restatement of the reference's TF-1.4 graph (oracle/, "port") timed on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F32_PEAK_TFS = 157.3  # f32-input MFMA peak

WORKLOAD = dict(model="deepfm", field_size=39, feature_size=1_000_000, embedding_size=16, batch=4096,
                deep_layers=(400, 400, 400), dropout=(0.5, 0.5, 0.5), l2_reg=1e-4, learning_rate=5e-4, optimizer="Adam")


def cpu_baseline(w, seconds_budget=18.0):
    """The oracle (torch-CPU restatement of the TF-1.4 graph, NOT TF) timed on the host cores: same model, same batch
    shape, dense Adam over the full tables, on a bounded number of steps.  The thread count is the best of a short
    sweep (more threads than ~16-32 make the memory-bound dense-table passes slower on a 256-core host)."""
    import torch
    from oracle import deepctr_oracle as O
    cores = os.cpu_count() or 1
    cfg = O.Config(model=w["model"], field_size=w["field_size"], feature_size=w["feature_size"],
                   embedding_size=w["embedding_size"], deep_layers=w["deep_layers"], dropout=w["dropout"],
                   l2_reg=w["l2_reg"], learning_rate=w["learning_rate"], optimizer=w["optimizer"])
    p = O.init_params(cfg, seed=1, scale=0.01)
    opt = O.Optimizer(cfg, p)
    B = w["batch"]
    batches = [O.synth_batch(B, cfg.field_size, cfg.feature_size, seed=20260924 + i) for i in range(4)]
    best_nt, best_t = 1, float("inf")
    for nt in sorted({min(cores, n) for n in (8, 16, 32)}):
        torch.set_num_threads(nt)
        O.train_step(cfg, p, opt, *batches[0])
        t0 = time.perf_counter()
        for i in range(2):
            O.train_step(cfg, p, opt, *batches[i % 4])
        t = (time.perf_counter() - t0) / 2
        if t < best_t:
            best_nt, best_t = nt, t
    torch.set_num_threads(best_nt)
    t0 = time.perf_counter()
    n = 0
    while True:
        O.train_step(cfg, p, opt, *batches[n % 4])
        n += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or n >= 400:
            break
    return {"value": round(B * n / el, 1), "unit": "examples/sec", "cores": best_nt, "kind": "port",
            "sample": "%d train steps of the same workload (batch %d), torch-CPU fp32 restatement of the TF-1.4 graph (dense table "
                      "gradient + dense Adam over all rows), %d threads = best of {8,16,32} on a %d-core host" % (n, B, best_nt, cores)}


def pmc_traffic_bytes(kernel_substr):
    """HBM bytes per launch of a kernel from the committed PMC summary (collected by tools/profile_round.sh in separate
    --pmc passes; a live bench run cannot sample counters).  None when the summary is absent."""
    import re
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.txt")
    if not os.path.exists(path):
        return None
    f = w = 0.0
    for line in open(path):
        if kernel_substr in line:
            mf = re.search(r"FETCH_SIZE=([0-9.e+]+)", line)
            mw = re.search(r"WRITE_SIZE=([0-9.e+]+)", line)
            if mf and mw:
                f += float(mf.group(1))
                w += float(mw.group(1))
    return int((2 * f + w) * 1024) if (f or w) else None


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
    ap.add_argument("--table-mode", default="dense_exact", choices=["dense_exact", "touched_rows"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--uniform-ids", action="store_true", help="uniform ids instead of Zipf (cache-worst case)")
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
    w = dict(WORKLOAD)
    B = w["batch"]

    sharded = world > 1 or bool(os.environ.get("DCTR_FORCE_SHARDED"))     # the env var exercises the RCCL path on one GPU
    if sharded:
        import torch.distributed as dist
        from tf_repos_amd.distributed import ShardedTrainer
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        try:
            trainer = ShardedTrainer(w, rank, world, dev, table_mode=args.table_mode)        # native C++ driver over RCCL
        except Exception as e:                                                                # noqa: BLE001
            # the same row-sharded protocol orchestrated from Python through torch.distributed (slower host side, same results)
            print("rank %d: native sharded driver unavailable (%s); using the torch.distributed driver" % (rank, e), file=sys.stderr, flush=True)
            trainer = ShardedTrainer(w, rank, world, dev, table_mode=args.table_mode, driver="python")
        step = lambda i, v, l, nxt: trainer.train_step(i, v, l, next_ids=nxt)      # routes the next batch's ids a step ahead
        barrier = dist.barrier
    else:
        eng = Engine(EngineConfig(model=w["model"], field_size=w["field_size"], feature_size=w["feature_size"],
                                  embedding_size=w["embedding_size"], deep_layers=w["deep_layers"], dropout=w["dropout"],
                                  l2_reg=w["l2_reg"], learning_rate=w["learning_rate"], optimizer=w["optimizer"],
                                  table_mode=args.table_mode, max_batch=B, seed=1,
                                  use_graph=os.environ.get("DCTR_USE_GRAPH", "0") == "1"))
        rng = np.random.default_rng(1)
        for name, shp in eng.param_shapes.items():
            eng.set_param(name, rng.normal(0, 0.01, size=shp).astype(np.float32))
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
    for i in range(nb):
        ids, vals, labels = synth_batch(B, w["field_size"], w["feature_size"], seed=20260924 + 1 + rank * 1000 + i,
                                        uniform_ids=args.uniform_ids)
        t = (torch.from_numpy(ids).to(dev), torch.from_numpy(vals).to(dev), torch.from_numpy(labels).to(dev))
        # resident inputs: the 8 synthetic batches live in the engine's 8 input slots (what the input pipeline's H2D copy
        # targets), so a step reads them in place
        si, sv, sl = (trainer.eng if sharded else eng).input_slot(i)
        si[:B].copy_(t[0]); sv[:B].copy_(t[1]); sl[:B].copy_(t[2])
        t = (si[:B], sv[:B], sl[:B])
        batches.append(t)

    for s in range(args.warmup):
        step(*batches[s % nb], batches[(s + 1) % nb][0])
    barrier()
    torch.cuda.synchronize()
    (trainer.eng if sharded else eng).step_timer(True)          # hipEvents around the roofline kernel inside the timed steps
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(*batches[(args.warmup + s) % nb], batches[(args.warmup + s + 1) % nb][0])
    t_enq = time.perf_counter() - t0          # host time to ENQUEUE the steps (the GPU may still be running)
    barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    in_ms, in_n = (trainer.eng if sharded else eng).step_timer(False)
    if sharded:
        import torch.distributed as dist
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    out = None
    if rank == 0:
        out = {
            "metric": "examples/sec DeepFM Criteo-39-field batch 4096", "value": round(B * world * args.steps / el, 1),
            "unit": "examples/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * el / args.steps, 4), "host_enqueue_ms_per_step": round(1e3 * t_enq / args.steps, 4),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic Criteo-shaped (Zipf categorical ids), random-init weights",
            "config": {"workload": "DeepFM 39 fields, vocab 1e6, emb_dim 16, batch 4096/GPU, MLP 400-400-400 keep 0.5, Adam "
                                   "(BASELINE configs[1])", "global_batch": B * world, "table_mode": args.table_mode,
                       "parallelism": "single GPU" if world == 1 else "row-sharded tables (id %% %d) + data-parallel dense" % world,
                       "ids": "uniform" if args.uniform_ids else "zipf"},
        }
    if rank == 0:
        # ---- per-stage timing (hipEvents around a graph of back-to-back launches, on torch's current stream)
        F, K, V = w["field_size"], w["embedding_size"], w["feature_size"]
        e = trainer.eng if sharded else eng
        rows = (V + world - 1) // world                            # rows of this rank's table shard
        names = ["opt_table", "mlp0_fwd", "mlp0_dgrad", "mlp0_wgrad"]
        if not sharded:
            names = ["embed_gather", "forward", "head", "backward_dense", "group_ids", "scatter", "opt_dense"] + names
        stages = {name: e.time_stage(name, iters=30) for name in names}
        gather_bytes = B * (F * (12 + 8 * K) + 8)                 # algorithmic bytes of the gather
        # dense-exact table step as implemented: theta,m,v read + write (6 streams) + the 4-byte slot word per row; the per-row
        # gradient is NOT a dense stream here (only the ~U touched rows read a compact gradient row), so SURVEY 8d's 7-stream
        # figure (7*V*(K+1)*4 = 476 MB) would flatter the kernel -- 6 streams (412 MB) is what the algorithm must move
        table_bytes = 6 * rows * (K + 1) * 4 + 4 * rows
        mlp0_flops = 2.0 * B * (F * K) * w["deep_layers"][0]
        kernels = {
            "opt_table_dense_adam": {"bound": "hbm", "ms": stages["opt_table"], "achieved": table_bytes / stages["opt_table"] / 1e6,
                                     "peak": HBM_PEAK_GBS, "unit": "GB/s"},
            "mlp0_fwd_gemm": {"bound": "mfma", "ms": stages["mlp0_fwd"], "achieved": mlp0_flops / stages["mlp0_fwd"] / 1e9,
                              "peak": MFMA_F32_PEAK_TFS, "unit": "TFLOP/s"},
            "mlp0_dgrad_gemm": {"bound": "mfma", "ms": stages["mlp0_dgrad"], "achieved": mlp0_flops / stages["mlp0_dgrad"] / 1e9,
                                "peak": MFMA_F32_PEAK_TFS, "unit": "TFLOP/s"},
            "mlp0_wgrad_gemm": {"bound": "mfma", "ms": stages["mlp0_wgrad"], "achieved": mlp0_flops / stages["mlp0_wgrad"] / 1e9,
                                "peak": MFMA_F32_PEAK_TFS, "unit": "TFLOP/s"},
        }
        if not sharded:
            kernels["embed_gather_fwd"] = {"bound": "hbm", "ms": stages["embed_gather"], "achieved": gather_bytes / stages["embed_gather"] / 1e6,
                                           "peak": HBM_PEAK_GBS, "unit": "GB/s"}
        copy_gbps = e.measure_copy_bandwidth(1 << 30, 20)          # this box's measured HBM roofline (1 GiB float4 copy, read + write)
        out["hbm_measured_copy_GBps"] = round(copy_gbps, 1)
        for k in kernels.values():
            if k["bound"] == "hbm":
                k["frac_of_measured_copy"] = round(k["achieved"] / copy_gbps, 4)
            k["frac"] = round(k["achieved"] / k["peak"], 4)
            k["achieved"] = round(k["achieved"], 2)
            k["ms"] = round(k["ms"], 5)
        # the dominant kernel family of the step is the MLP GEMM (half of the kernel time; the dense-exact table pass, once the
        # longest single launch, now runs as a background kernel under the GEMMs).  `roofline` is the first layer's forward GEMM
        # (4096 x 624 x 400) AS IT RUNS IN THE TIMED STEPS: hipEvents on the step's stream around that launch (dctr_step_timer);
        # `kernels` holds the same kernels timed alone, back to back.
        dom = "mlp0_fwd_gemm"
        r = dict(kernels[dom])
        r["kernel"] = dom + " (gemm_f32_mfma<true,true,1>, layer 0: 4096x624x400)"
        if in_n > 0:
            r["ms_alone"] = r["ms"]
            r["ms"] = round(in_ms, 5)
            r["achieved"] = round(mlp0_flops / in_ms / 1e9, 2)
            r["frac"] = round(r["achieved"] / r["peak"], 4)
            r["launches_timed"] = in_n
        r["traffic"] = None
        r["hbm_kernel"] = dict(kernels["opt_table_dense_adam"], traffic=pmc_traffic_bytes("opt_table_kernel") if not sharded else None)
        r["traffic_source"] = "profiles/r01_pmc_traffic.txt: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE counts 1/2 of 16-B/lane streams; MI355X_MICROARCH.md HBM section)"
        out["roofline"] = r
        out["kernels"] = kernels
        out["stage_ms"] = {k: round(v, 5) for k, v in stages.items()}
        if not sharded and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w)
            out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    done.set()
    if sharded:
        import torch.distributed as dist
        trainer.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
