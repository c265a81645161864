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


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _headers_mtime():
    m = os.path.getmtime(os.path.join(INCLUDE, "deepctr_hip.h"))
    for f in os.listdir(CSRC):
        if f.endswith(".h"):
            m = max(m, os.path.getmtime(os.path.join(CSRC, f)))
    return m


def _compile(src: str, hdr_m: float, force: bool, verbose: bool) -> str:
    obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(path), hdr_m)):
        return obj
    cmd = [HIPCC] + FLAGS + ["-x", "hip", "-c", path, "-o", obj]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stdout))
    if verbose and r.stdout.strip():
        print(r.stdout)
    return obj


def build_library(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = _sources()
    hdr_m = _headers_mtime()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, hdr_m, force, verbose), srcs))
    if (force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs)):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lpthread"]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stdout)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
