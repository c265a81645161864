"""tools/prof_summary.py `timeline`: which traced step the committed `profiles/*_step_timeline.txt` shows.  Under rocprofv3 every HIP
call costs more and on some boxes the enqueueing thread becomes the bottleneck for part of the run (steps with idle gaps); the tool
therefore shows the FASTEST anchor-to-anchor interval among the steps of the usual kernel count and prints the median / slowest beside
it -- and it must not mistake the bench's back-to-back single-kernel timing loops (hundreds of one-kernel "steps") for training steps."""
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_timeline_picks_the_least_disturbed_training_step(tmp_path):
    db = tmp_path / "trace.db"
    con = sqlite3.connect(str(db))
    cur = con.cursor()
    cur.execute("create table kernels (name text, start integer, end integer, queue_id integer)")
    names = ["void dctr::gather_fwd_kernel<4, 8, 1, 5, true>(x)"] + ["k%d" % i for i in range(22)]
    t = 0
    for step in range(60):                      # 23-kernel steps; two of three carry a 50-us hole (a host-bound stretch)
        hole = 50_000 if step % 3 else 0
        for n in names:
            cur.execute("insert into kernels values (?,?,?,?)", (n, t, t + 10_000, 1))
            t += 10_000 + (hole if n == "k5" else 0)
    for _ in range(500):                        # the bench's stage timing: the gather alone, back to back
        cur.execute("insert into kernels values (?,?,?,?)", ("void dctr::gather_fwd_kernel<4, 8, 1, 5, false>(x)", t, t + 5_000, 1))
        t += 5_500
    con.commit(); con.close()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_summary.py"), "timeline", str(db)], capture_output=True, text=True, check=True).stdout
    head = out.splitlines()[0]
    assert "23 kernels" in head and "gather to next gather 230.0 us" in head, head
    assert "fastest of" in head and "median 280.0 us" in head, head
    assert out.count("gather_fwd_kernel") >= 2          # two consecutive steps are listed
