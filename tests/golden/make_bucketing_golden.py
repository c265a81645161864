"""Generates the integer-bucketing golden fixtures by RUNNING THE REFERENCE ITSELF in this container:
/root/reference/deep_ctr/Feature_pipeline/get_criteo_feature.py (pure Python, runs unmodified under python3).

    python tests/golden/make_bucketing_golden.py

Writes tests/golden/criteo_small/{train.txt,test.txt} (synthetic Criteo-format TSV, fixed seed) and the reference's
outputs {tr.libsvm,va.libsvm,te.libsvm,feature_map}.  The GPU box has no /root/reference: tests only read the
committed fixtures.
"""
import os
import random
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "criteo_small")
REF = "/root/reference/deep_ctr/Feature_pipeline/get_criteo_feature.py"


def synth_tsv(path, n, with_label, seed):
    rng = random.Random(seed)
    vocab = [["%08x" % rng.getrandbits(32) for _ in range(rng.randint(3, 40))] for _ in range(26)]
    with open(path, "w") as f:
        for _ in range(n):
            cols = []
            if with_label:
                cols.append(str(int(rng.random() < 0.25)))
            for i in range(13):       # integer features, some empty, some above the clip point
                r = rng.random()
                cols.append("" if r < 0.1 else str(int(rng.expovariate(1.0 / (5 + 40 * i)))))
            for c in range(26):       # categorical features: Zipf-ish draws, some empty
                r = rng.random()
                if r < 0.05:
                    cols.append("")
                else:
                    v = vocab[c]
                    cols.append(v[min(int(rng.paretovariate(1.2)) - 1, len(v) - 1)])
            f.write("\t".join(cols) + "\n")


def main():
    os.makedirs(OUT, exist_ok=True)
    synth_tsv(os.path.join(OUT, "train.txt"), 400, True, 20260924)
    synth_tsv(os.path.join(OUT, "test.txt"), 60, False, 20260925)
    subprocess.check_call([sys.executable, REF, "--input_dir=" + OUT + "/", "--output_dir=" + OUT + "/", "--cutoff=3"])
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
