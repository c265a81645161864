#!/usr/bin/env python
"""Trains the reference's OWN deep_ctr/Model_pipeline scripts on the MI355X (runs on the GPU box).

/root/reference does not exist there, and reference sources are never committed here, so the caller stages them first into the
git-ignored scratch directory .ref_stage/ (it travels with the gpurun snapshot):

    mkdir -p .ref_stage && cp /root/reference/deep_ctr/Model_pipeline/{DeepFM,DCN,PNN,NFM,AFM,DeepMVM,wide_n_deep}.py .ref_stage/
    gpurun -- 'python tools/reference_scripts_gpu.py > gpurun_out/r02_reference_scripts.txt 2>&1'

Part 1 (the scripts as a user runs them, deep_ctr/README.md:49 flags): `python -m tf_repos_amd.run_reference <script>
--task_type=train|eval|infer ...` on synthetic Criteo-shaped libsvm files; the log keeps the Estimator's own loss / examples/sec
lines, the eval AUC and the head of pred.txt.
Part 2 (parity): the same model_fn / input_fn objects driven through Estimator.train with keep_prob 1.0, the final variables
compared with oracle/deepctr_oracle.py trained on the same file from the same initial values.
"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STAGE = os.path.join(ROOT, ".ref_stage")
DATA = "/tmp/ref_data"
F, V = 39, 117581


def synth_lines(n, seed):
    """Criteo-shaped libsvm text: fields 1-13 numeric (id = field index, value %.6f), 14-39 categorical (Zipf rank inside the
    field's slice of the vocabulary, value 1), label ~ Bernoulli(0.25) with a planted signal so that AUC moves."""
    rng = np.random.default_rng(seed)
    span = (V - 14) // 26
    w = np.random.default_rng(99).normal(0, 1, size=V)
    lines = []
    for _ in range(n):
        ids = list(range(1, 14)) + [14 + f * span + min(int(rng.zipf(1.2)) - 1, span - 1) for f in range(26)]
        vals = [round(float(rng.random()), 6) for _ in range(13)] + [1.0] * 26
        s = sum(w[i] * v for i, v in zip(ids, vals)) * 0.35 - 1.2
        y = int(rng.random() < 1.0 / (1.0 + np.exp(-s)))
        lines.append("%d " % y + " ".join("%d:%s" % (i, ("%.6f" % v).rstrip("0").rstrip(".") if v != 1.0 else "1") for i, v in zip(ids, vals)))
    return "\n".join(lines) + "\n"


def make_data():
    os.makedirs(DATA, exist_ok=True)
    for name, n, seed in (("tr.libsvm", 65536, 1), ("va.libsvm", 8192, 2), ("te.libsvm", 4096, 3)):
        with open(os.path.join(DATA, name), "w") as f:
            f.write(synth_lines(n, seed))


def run(script, extra, task, model_dir):
    cmd = [sys.executable, "-m", "tf_repos_amd.run_reference", os.path.join(STAGE, script), "--task_type=" + task,
           "--model_dir=" + model_dir, "--data_dir=" + DATA, "--dt_dir=r02", "--log_steps=50", "--num_threads=8",
           "--field_size=39", "--feature_size=%d" % V, "--batch_size=256", "--num_epochs=1"] + extra
    t0 = time.time()
    r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    keep = [ln for ln in r.stdout.splitlines() if any(k in ln for k in ("loss", "auc", "examples/sec", "global_step", "Error", "error", "Traceback", "Saving", "Restor"))]
    print("$ " + " ".join(cmd[2:]))
    print("\n".join("    " + ln[:220] for ln in keep[-14:]))
    print("    [rc %d, %.1f s]" % (r.returncode, time.time() - t0), flush=True)
    return r.returncode


# deep_ctr/README.md:49 for DeepFM; deep_ctr/run.sh:11-20 for the others (embedding sizes kept small enough for a quick run)
RUNS = [
    ("DeepFM.py", ["--learning_rate=0.0005", "--optimizer=Adam", "--embedding_size=8", "--deep_layers=400,400,400", "--dropout=0.5,0.5,0.5"]),
    ("DCN.py", ["--learning_rate=0.0005", "--optimizer=Adam", "--embedding_size=8", "--deep_layers=400,400", "--dropout=0.5,0.5", "--cross_layers=3"]),
    ("PNN.py", ["--learning_rate=0.0005", "--optimizer=Adam", "--embedding_size=8", "--deep_layers=256,128", "--dropout=0.5,0.5", "--model_type=Inner"]),
    ("NFM.py", ["--learning_rate=0.005", "--optimizer=Adam", "--embedding_size=16", "--deep_layers=128,64", "--dropout=0.5,0.8,0.8", "--l2_reg=0.001"]),
    ("AFM.py", ["--learning_rate=0.01", "--optimizer=Adam", "--embedding_size=16", "--attention_layers=16", "--dropout=1.0,0.5", "--l2_reg=0.001"]),
    ("DeepMVM.py", ["--learning_rate=0.0005", "--optimizer=Adam", "--embedding_size=8", "--deep_layers=400,400", "--dropout=0.5,0.5"]),
]


def part1():
    fails = 0
    for script, extra in RUNS:
        if not os.path.exists(os.path.join(STAGE, script)):
            print("(not staged: %s)" % script)
            continue
        md = "/tmp/ref_model/%s/" % script[:-3]
        print("=" * 30, script)
        for task in ("train", "eval", "infer"):
            fails += run(script, extra + (["--clear_existing_model=True"] if task == "train" else []), task, md) != 0
        pred = os.path.join(DATA, "pred.txt")
        if os.path.exists(pred):
            head = open(pred).read().split("\n")[:5]
            print("    pred.txt (%d lines): %s" % (sum(1 for _ in open(pred)), " ".join(head)))
            os.remove(pred)
    return fails


def part2():
    """model_fn / input_fn of the staged scripts through Estimator.train, final variables == oracle (keep_prob 1)."""
    import torch
    import tf_repos_amd.tf_shim as shim
    from oracle import deepctr_oracle as O
    from tf_repos_amd.run_reference import load_reference_module
    small = os.path.join(DATA, "tr_small.libsvm")
    with open(small, "w") as f:
        f.write(synth_lines(5 * 256 + 17, 11))            # a ragged last batch
    worst_all = 0.0
    for script, flags, P in [
        ("DeepFM.py", {}, dict(deep_layers="32,16", dropout="1.0,1.0")),
        ("DCN.py", {}, dict(deep_layers="32,16", dropout="1.0,1.0", cross_layers=2)),
        ("PNN.py", {"model_type": "Inner"}, dict(deep_layers="32,16", dropout="1.0,1.0")),
        ("NFM.py", {}, dict(deep_layers="32,16", dropout="1.0,1.0,1.0")),
        ("AFM.py", {}, dict(attention_layers="8", dropout="1.0,1.0")),
        ("DeepMVM.py", {}, dict(deep_layers="32,16", dropout="1.0,1.0")),
    ]:
        if not os.path.exists(os.path.join(STAGE, script)):
            continue
        mod = load_reference_module(os.path.join(STAGE, script))
        for k, v in flags.items():
            setattr(shim.FLAGS_MODULE.FLAGS, k, v)
        tf = sys.modules["tensorflow"]
        params = dict(field_size=F, feature_size=V, embedding_size=8, learning_rate=0.01, batch_norm_decay=0.9, l2_reg=1e-3)
        params.update(P)
        md = "/tmp/ref_parity/%s" % script[:-3]
        subprocess.run(["rm", "-rf", md])
        est = tf.estimator.Estimator(model_fn=mod.model_fn, model_dir=md, params=params)
        tr_fn = lambda: mod.input_fn([small], num_epochs=1, batch_size=256)
        spec, lowered, pipe, variables = est._build(tr_fn, "train")
        est._ensure_engine(lowered, variables, 256)
        kw = lowered.config_kwargs
        ocfg = O.Config(**{k: kw[k] for k in ("model", "field_size", "feature_size", "embedding_size", "deep_layers", "dropout", "attention_layers",
                                              "cross_layers", "l2_reg", "learning_rate", "optimizer") if k in kw})
        p = {k: torch.from_numpy(est._engine.get_param(k).copy()) for k in O.param_shapes(ocfg)}
        t0 = time.time()
        est.train(input_fn=tr_fn)
        dt = time.time() - t0
        opt = O.Optimizer(ocfg, p)
        ids, vals, labels = O.parse_libsvm(open(small).read(), F)
        # AFM: the attention network sits behind a softmax over 741 pairs -- most of its gradient elements are ~1e-8, and Adam's
        # 1/sqrt(v) turns their fp32 rounding into a visible fraction of lr on either side.  As in the golden-fixture tests
        # (tests/test_model_golden.py var_err) the elements whose gradient was rounding noise at ANY step are left out of the
        # comparison -- element by element, not the whole model (round 3 divided AFM's error by 400)
        live = None
        for s in range(0, len(labels), 256):
            if lowered.model == "afm":
                _, g, _ = O.grads(ocfg, p, ids[s:s + 256], vals[s:s + 256], labels[s:s + 256], train=True)
                now = {k: (v.abs() >= 1e-4 * v.abs().max()).numpy() for k, v in g.items()}
                live = now if live is None else {k: live[k] & now[k] for k in now}
            O.train_step(ocfg, p, opt, ids[s:s + 256], vals[s:s + 256], labels[s:s + 256])

        def err(ename, tfname):
            d = np.abs(est.get_variable_value(tfname) - p[ename].numpy())
            if live is not None and ename in live:
                return float(d[live[ename]].max()) if live[ename].any() else 0.0
            return float(d.max())
        worst = max(err(ename, tfname) for ename, tfname in lowered.name_map.items())
        res = est.evaluate(input_fn=lambda: mod.input_fn([os.path.join(DATA, "va.libsvm")], num_epochs=1, batch_size=256))
        worst_all = max(worst_all, worst)
        print("parity %-10s model=%-6s steps=%d  max |variable - oracle| = %.2e  eval auc %.4f loss %.5f  (train %.2f s)" % (
            script, lowered.model, est._engine.global_step, worst, res["auc"], res["loss"], dt), flush=True)
        est._engine.close()
    return worst_all


if __name__ == "__main__":
    if not os.path.isdir(STAGE):
        raise SystemExit("stage the reference scripts into .ref_stage/ first (see the docstring)")
    make_data()
    f = part1()
    w = part2()
    print("part 1 failures: %d;  part 2 worst |variable - oracle| = %.2e" % (f, w))
    sys.exit(1 if (f or w > 2e-5) else 0)
