"""Summarises rocprofv3 outputs (rocpd sqlite: *_results.db) into the text files committed under profiles/.

    python tools/prof_summary.py stats  <results.db>            # per-kernel calls / total / average (like --stats)
    python tools/prof_summary.py pmc    <results.db> [...]      # per-kernel mean counter values (one db per --pmc pass)
"""
import collections
import sqlite3
import sys


def stats(path):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print("%-92s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "%"))
    for name, calls, tot, avg, pct in rows:
        print("%-92s %8d %14.0f %12.1f %7.2f" % (name[:92], calls, tot, avg, pct))


def pmc(paths):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in paths:
        cur = sqlite3.connect(path).cursor()
        cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
        ki, ci, vi = cols.index("kernel_name"), cols.index("counter_name"), cols.index("value")
        di = cols.index("dispatch_id") if "dispatch_id" in cols else None
        per = collections.defaultdict(float)
        for r in cur.execute("select * from counters_collection"):
            per[(r[ki], r[ci], r[di] if di is not None else 0)] += r[vi]      # sum over XCDs / instances of one dispatch
        for (k, c, _d), v in per.items():
            agg[k][c].append(v)
    print("%-80s %s" % ("kernel", "mean counter value per dispatch"))
    for k, d in sorted(agg.items()):
        print("%-80s %s" % (k[:80], "  ".join("%s=%.4g (n=%d)" % (c, sum(v) / len(v), len(v)) for c, v in sorted(d.items()))))


def timeline(path, anchor="gather_fwd_kernel", which=None):
    """Start offset / duration of every kernel between two consecutive launches of `anchor` (one training step).  Which step: the
    FASTEST (anchor to anchor) among the traced steps of the usual kernel count -- under the tracer every API call costs more and on
    some boxes the enqueueing thread becomes the bottleneck (gaps with nothing running); the fastest step is the one the tracer
    disturbed least.  The header says how the others did."""
    cur = sqlite3.connect(path).cursor()
    cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
    ni, si, ei = cols.index("name"), cols.index("start"), cols.index("end")
    qi = cols.index("queue_id") if "queue_id" in cols else (cols.index("stream_id") if "stream_id" in cols else None)
    rows = sorted(cur.execute("select * from kernels"), key=lambda r: r[si])
    marks = [i for i, r in enumerate(rows) if anchor in r[ni]]
    note = ""
    if which is None:
        # (the bench also times single kernels back to back -- hundreds of anchor-to-anchor intervals of ONE kernel: a step has >= 10)
        counts = collections.Counter(marks[j + 1] - marks[j] for j in range(len(marks) - 2) if marks[j + 1] - marks[j] >= 10)
        usual = counts.most_common(1)[0][0] if counts else 0
        cand = [j for j in range(5, len(marks) - 2) if marks[j + 1] - marks[j] == usual and marks[j + 2] - marks[j + 1] == usual]
        if cand:
            dur = {j: rows[marks[j + 1]][si] - rows[marks[j]][si] for j in cand}
            which = min(cand, key=lambda j: dur[j])
            ds = sorted(dur.values())
            note = " [fastest of %d traced steps of %d kernels; median %.1f us, slowest %.1f us]" % (len(cand), usual, ds[len(ds) // 2] / 1e3, ds[-1] / 1e3)
        else:
            which = 40
    if len(marks) < which + 2:
        which = max(0, len(marks) - 2)
    a, b = marks[which], marks[which + 1]
    c = marks[which + 2] if which + 2 < len(marks) else b
    t0 = rows[a][si]
    print("one step: %d kernels, %.1f us from first start to last end; gather to next gather %.1f us%s" % (
        b - a, (max(r[ei] for r in rows[a:b]) - t0) / 1e3, (rows[b][si] - t0) / 1e3, note))
    print("%10s %10s %10s %6s  %s" % ("start_us", "dur_us", "end_us", "queue", "kernel"))
    for r in rows[a:c]:           # two consecutive steps: what trails a step runs under the head of the next
        print("%10.1f %10.1f %10.1f %6s  %s" % ((r[si] - t0) / 1e3, (r[ei] - r[si]) / 1e3, (r[ei] - t0) / 1e3, r[qi] if qi is not None else "-", r[ni][:70]))


if __name__ == "__main__":
    if sys.argv[1] == "timeline":
        timeline(sys.argv[2])
        sys.exit(0)
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
