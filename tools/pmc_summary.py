"""Summarise a rocprofv3 --pmc run (sqlite rocpd db or csv) per kernel: mean counter values."""
import sqlite3, sys, glob, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
if 'counters_collection' in tabs:
    cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
    print(cols)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    ki = cols.index('kernel_name') if 'kernel_name' in cols else None
    ci = cols.index('counter_name'); vi = cols.index('value')
    for r in cur.execute("select * from counters_collection"):
        agg[r[ki][:60]][r[ci]].append(r[vi])
    for k, d in agg.items():
        print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
else:
    print([t for t in tabs if 'counter' in t.lower() or 'pmc' in t.lower()])
