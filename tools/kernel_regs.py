#!/usr/bin/env python
"""Per-kernel register / LDS / scratch usage of the built objects (AMDGPU metadata notes of the gfx950 code objects):
    python tools/kernel_regs.py [substring ...]      # e.g. gemm_dr lag_advance scatter_apply
Reads tf_repos_amd/_lib/obj/*.o; each host object embeds the device code object as a clang offload bundle."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "tf_repos_amd", "_lib", "obj")
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels_of(obj):
    with tempfile.TemporaryDirectory() as td:
        import shutil
        cp = os.path.join(td, "x.o")
        shutil.copy(obj, cp)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", cp], capture_output=True, text=True, cwd=td)
        cos = [f for f in os.listdir(td) if "gfx950" in f]
        if not cos:
            return []
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(td, cos[0])], capture_output=True, text=True).stdout
    out = []
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        blk = ".agpr_count:" + blk
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
        out.append(dict(name=g("name"), vgpr=g("vgpr_count"), agpr=g("agpr_count"), sgpr=g("sgpr_count"), lds=g("group_segment_fixed_size"),
                        scratch=g("private_segment_fixed_size"), spill=g("vgpr_spill_count")))
    return out


def main():
    pats = sys.argv[1:]
    for f in sorted(os.listdir(OBJ)):
        if not f.endswith(".o"):
            continue
        for k in kernels_of(os.path.join(OBJ, f)):
            dem = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip()
            if pats and not any(p in dem for p in pats):
                continue
            print("%-14s vgpr %3s agpr %3s sgpr %3s lds %6s scratch %4s spill %3s  %s" % (f, k["vgpr"], k["agpr"], k["sgpr"], k["lds"], k["scratch"], k["spill"], dem[:150]))


if __name__ == "__main__":
    main()
