"""Where the GPU idles inside ONE steady-state training step of a rocprofv3 results .db: the union of all kernel intervals between the
last two launches of a marker kernel, the idle time, and the largest idle gaps with the kernels on either side.
    python tools/prof_gaps.py /tmp/prof_train/t_results.db k_fused_adam [top]"""
import sqlite3
import sys

db, marker = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
c = sqlite3.connect(db)
marks = c.execute("select start from kernels where name like ? order by start", ("%" + marker + "%",)).fetchall()
t0, t1 = marks[-2][0], marks[-1][0]
rows = c.execute("select name, start, end from kernels where start > ? and start <= ? order by start", (t0, t1)).fetchall()
busy, gaps, cur_end, cur_name = 0, [], rows[0][1], "(step start)"
for name, s, e in rows:
    if s > cur_end:
        gaps.append((s - cur_end, cur_name, name, cur_end - t0))
        busy_from = s
    else:
        busy_from = cur_end
    if e > cur_end:
        busy += e - busy_from
        cur_end, cur_name = e, name
wall = t1 - t0
print(f"step {wall / 1e6:.2f} ms wall, GPU busy (union of kernels) {busy / 1e6:.2f} ms, idle {(wall - busy) / 1e6:.2f} ms in {len(gaps)} gaps; "
      f"sum of kernel durations {sum(e - s for _, s, e in rows) / 1e6:.2f} ms")
# idle time by 10 % slices of the step
sl = [0.0] * 10
for g, _, _, at in gaps:
    sl[min(9, int(10 * at / wall))] += g
print("idle ms per tenth of the step:", " ".join(f"{v / 1e6:.1f}" for v in sl))
for g, a, b, at in sorted(gaps, reverse=True)[:top]:
    print(f"{g / 1e3:9.1f} us at {at / 1e6:7.2f} ms  after {a[:60]:60s} before {b[:60]}")
