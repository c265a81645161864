"""Every DCTR_* environment knob the sources read, with where it is read and the comment that stands at (or right above) that line.
The knobs are A/B switches left from measured experiments plus a few run-time selectors; none is needed to use the
library.  usage: python tools/list_knobs.py > tools/KNOBS.txt"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAT = re.compile(r'(?:getenv\(|environ(?:\.get)?[\(\[]\s*)"(DCTR_[A-Z0-9_]+)"')
files = []
for d, _dirs, fs in os.walk(ROOT):
    if any(p in d for p in (".git", "gpurun_out", "_lib", "__pycache__", ".ref_stage")):
        continue
    for f in fs:
        if f.endswith((".hip", ".h", ".py", ".sh")) and f != "list_knobs.py":
            files.append(os.path.join(d, f))
seen = {}
for path in sorted(files):
    lines = open(path, errors="replace").read().split("\n")
    for i, line in enumerate(lines):
        for m in PAT.finditer(line):
            name = m.group(1)
            note = ""
            c = line.find("//") if path.endswith((".hip", ".h")) else line.find("#")
            if c >= 0 and c > m.start():
                note = line[c:].lstrip("/# ").strip()
            if not note:            # the comment block right above
                j = i - 1
                block = []
                while j >= 0 and lines[j].strip().startswith(("//", "#")):
                    block.insert(0, lines[j].strip().lstrip("/# ").strip())
                    j -= 1
                note = " ".join(block)[-400:]
            entry = (os.path.relpath(path, ROOT), i + 1, note)
            if entry not in seen.setdefault(name, []):
                seen[name].append(entry)
for name in sorted(seen):
    where = seen[name]
    print(name)
    for rel, ln, note in where[:3]:
        print("    %s:%d  %s" % (rel, ln, note[:300]))
    if len(where) > 3:
        print("    (+ %d more places)" % (len(where) - 3))
print("\n%d knobs" % len(seen), file=sys.stderr)
