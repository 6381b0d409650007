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
if len(sys.argv) > 3:   # timeline of the last N kernels
    n = int(sys.argv[3])
    last = c.execute("select name, start, end, stream_id, queue_id from kernels order by start desc limit ?", (n,)).fetchall()[::-1]
    t0 = last[0][1]
    for name, s, e, st, q in last:
        print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} q{q} {name[:70]}")
