"""Builds libdeepctr_hip.so in-tree with hipcc for gfx950 (no cmake, no JIT cache).

`python -m tf_repos_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles without a GPU.
The .so lands in tf_repos_amd/_lib/ (git-ignored, but it travels with a gpurun snapshot).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "_lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libdeepctr_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
         "-I", INCLUDE]


# per-file flags.  gemm_dr3.hip, gemm_ts.hip: hipcc's SLP pass pairs the split's f32 subtractions into v_pk_add_f32, slower beside MFMAs than two
# v_sub_f32 (see the file's header)
FILE_FLAGS = {"gemm_dr3.hip": ["-fno-slp-vectorize"], "gemm_ts.hip": ["-fno-slp-vectorize"]}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def sources_hash() -> str:
    """sha256 over the library's sources (csrc/*, include/deepctr_hip.h): names a build independently of when it was made -- what the
    committed profiles are stamped with (tools/profile_round.sh) and bench.py compares against."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".cpp", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(INCLUDE, "deepctr_hip.h"), "rb").read())
    return h.hexdigest()


def _headers_mtime():
    m = os.path.getmtime(os.path.join(INCLUDE, "deepctr_hip.h"))
    for f in os.listdir(CSRC):
        if f.endswith(".h"):
            m = max(m, os.path.getmtime(os.path.join(CSRC, f)))
    return m


def _compile(src: str, hdr_m: float, force: bool, verbose: bool, objdir: str = OBJDIR, extra=()) -> str:
    obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(path), hdr_m)):
        return obj
    cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(src, []) + list(extra) + ["-x", "hip", "-c", path, "-o", obj]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stdout))
    if verbose and r.stdout.strip():
        print(r.stdout)
    return obj


def build_library(force: bool = False, verbose: bool = True, variant: str = "", extra_flags=()) -> str:
    """variant: an experimental build beside the product library (A/B runs on the GPU box): objects in _lib/obj_<variant>, library
    _lib/libdeepctr_hip_<variant>.so, compiled with extra_flags (e.g. -DNAME); selected at run time with DCTR_LIB_VARIANT=<variant>."""
    objdir = OBJDIR + ("_" + variant if variant else "")
    lib = LIB if not variant else os.path.join(LIBDIR, "libdeepctr_hip_%s.so" % variant)
    os.makedirs(objdir, exist_ok=True)
    srcs = _sources()
    hdr_m = _headers_mtime()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, hdr_m, force, verbose, objdir, extra_flags), srcs))
    if (force or not os.path.exists(lib) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in objs)):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-lpthread"]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stdout)
    return lib


if __name__ == "__main__":
    # python -m tf_repos_amd.build [--force] [--variant NAME -DFLAG ...]
    args = [a for a in sys.argv[1:] if a != "--force"]
    variant = ""
    if "--variant" in args:
        i = args.index("--variant")
        variant = args[i + 1]
        del args[i:i + 2]
    print(build_library(force="--force" in sys.argv, variant=variant, extra_flags=tuple(args)))
