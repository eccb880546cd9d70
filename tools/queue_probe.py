"""Which hardware queues do the dispatches of a rocprofv3 --kernel-trace run use, phase by phase?   python tools/queue_probe.py <db> [phases]
Splits the trace's time span into `phases` equal parts and prints, per part, the number of scoring-kernel dispatches per queue id
(and per stream id when the view has one)."""
import sqlite3
import sys

db = sys.argv[1]
phases = int(sys.argv[2]) if len(sys.argv) > 2 else 7
cur = sqlite3.connect(db).cursor()
cols = [r[1] for r in cur.execute("PRAGMA table_info(kernels)").fetchall()]
print("columns of `kernels`:", cols)
qcol = "queue_id" if "queue_id" in cols else None
scol = "stream_id" if "stream_id" in cols else None
sel = ", ".join(c for c in (qcol, scol) if c)
if not sel:
    sys.exit("no queue / stream columns")
rows = cur.execute(f"select start, name, {sel} from kernels where name like '%score_exact%' order by start").fetchall()
t0, t1 = rows[0][0], rows[-1][0]
for ph in range(phases):
    lo, hi = t0 + (t1 - t0) * ph / phases, t0 + (t1 - t0) * (ph + 1) / phases
    cnt = {}
    for r in rows:
        if lo <= r[0] < hi:
            cnt[r[2:]] = cnt.get(r[2:], 0) + 1
    print(f"phase {ph}: " + "  ".join(f"{k}:{v}" for k, v in sorted(cnt.items())))
