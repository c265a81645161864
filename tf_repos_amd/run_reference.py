"""Runs one of the reference's deep_ctr/Model_pipeline/*.py scripts UNCHANGED AT ITS tf.* CALL SITES on the MI355X engine:

    python -m tf_repos_amd.run_reference /path/to/deep_ctr/Model_pipeline/DeepFM.py --task_type=train --data_dir=... \
        --field_size=39 --feature_size=117581 ...

The scripts are Python-2.7 sources (DeepMTL/README.md:36); four mechanical py2->py3 fixes are applied to the source
text in memory before exec (the file on disk is never modified or copied): tabs -> 8 spaces (PNN.py mixes them),
`map(int|float, ...)` -> `list(map(...))` (DeepFM.py:110-111 index the result), `/ 2` -> `// 2` in the pair-count
expressions (PNN.py:113, AFM.py:132), `.iteritems()` -> `.items()` (Feature_pipeline/get_tfrecord.py); `np.int` / `np.float`
(removed from numpy 1.24) are aliased to the builtins for the feature scripts."""
import re
import sys


def py3_source(src: str) -> str:
    out = []
    for line in src.expandtabs(8).split("\n"):
        m = re.search(r"=\s*map\((int|float),", line)
        if m and "list(map" not in line:
            code = line.rstrip()
            line = code.replace("map(", "list(map(", 1) + ")"
        line = re.sub(r"\(field_size\s*-\s*1\)\s*/\s*2", "(field_size-1)//2", line)
        line = line.replace(".iteritems()", ".items()")          # Feature_pipeline/get_tfrecord.py:64,77,88
        out.append(line)
    return "\n".join(out)


def load_reference_module(path: str, name: str = "__reference__"):
    import types
    import tf_repos_amd.tf_shim as shim
    shim.install()
    shim.FLAGS_MODULE.FLAGS._reset()
    with open(path) as f:
        src = py3_source(f.read())
    import numpy as _np
    for _alias, _t in (("int", int), ("float", float)):            # np.int / np.float (get_tfrecord.py:72,84-85) left numpy in 1.24
        if not hasattr(_np, _alias):
            setattr(_np, _alias, _t)
    mod = types.ModuleType(name)
    mod.__file__ = path
    exec(compile(src, path, "exec"), mod.__dict__)
    return mod


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    path = sys.argv[1]
    sys.argv = [path] + sys.argv[2:]
    import tf_repos_amd.tf_shim as shim
    tf = shim.install()
    mod = load_reference_module(path, "__main__ref__")
    tf.logging.set_verbosity(tf.logging.INFO)
    tf.app.run(mod.main)


if __name__ == "__main__":
    main()
