"""Summarise a rocprofv3 results .db: per-kernel count / average / share; optional timeline of the last graph replay."""
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), avg(end-start), sum(end-start) from kernels group by name order by 4 desc").fetchall()
tot = sum(r[3] for r in rows)
print("name,calls,avg_us,share_pct")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"\"{r[0][:100]}\",{r[1]},{r[2] / 1e3:.2f},{100 * r[3] / tot:.2f}")
if len(sys.argv) > 3 and sys.argv[3] != "--step":   # timeline of the last N kernels
    n = int(sys.argv[3])
    last = c.execute("select name, start, end, stream_id, queue_id from kernels order by start desc limit ?", (n,)).fetchall()[::-1]
    t0 = last[0][1]
    for name, s, e, st, q in last:
        print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} q{q} {name[:70]}")


def last_interval(db, marker, top=60):
    """Per-kernel totals of ONE steady-state step: the kernels between the last two launches of `marker` (e.g. the optimiser
    kernel that ends a training step) — keeps the library warm-up (MIOpen's solver search) out of the table."""
    c = sqlite3.connect(db)
    marks = c.execute("select start from kernels where name like ? order by start", ("%" + marker + "%",)).fetchall()
    if len(marks) < 2:
        print("marker seen", len(marks), "times")
        return
    t0, t1 = marks[-2][0], marks[-1][0]
    rows = c.execute("select name, count(*), avg(end-start), sum(end-start) from kernels where start > ? and start <= ? group by name "
                     "order by 4 desc", (t0, t1)).fetchall()
    tot = sum(r[3] for r in rows)
    print(f"# one step = {(t1 - t0) / 1e6:.2f} ms wall between the last two '{marker}' launches; kernel time {tot / 1e6:.2f} ms")
    print("name,calls,avg_us,total_ms,share_pct")
    for r in rows[:top]:
        print(f"\"{r[0][:110]}\",{r[1]},{r[2] / 1e3:.2f},{r[3] / 1e6:.3f},{100 * r[3] / tot:.2f}")


if len(sys.argv) > 4 and sys.argv[3] == "--step":
    print()
    last_interval(db, sys.argv[4], int(sys.argv[2]))
