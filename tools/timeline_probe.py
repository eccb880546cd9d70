"""Development aid: from a rocprofv3 --kernel-trace database, how busy is the scoring kernel over the steady state of
a multi-stream run?   python tools/timeline_probe.py <trace_results.db>"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print("columns:", cols)
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
t_beg, t_end = min(r[1] for r in rows), max(r[2] for r in rows)
frac0, frac1 = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.4, 0.8)
win0, win1 = t_beg + frac0 * (t_end - t_beg), t_beg + frac1 * (t_end - t_beg)  # a window inside the run (ns)
rows = [r for r in rows if r[1] >= win0 and r[2] <= win1]
span = win1 - win0


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    return tot + (ce - cs if cs is not None else 0)


k4 = [(s, e) for n, s, e in rows if "score" in n]
oth = [(s, e) for n, s, e in rows if "score" not in n]
print(f"window {span / 1e6:.1f} ms: {len(k4)} scoring launches, sum {sum(e - s for s, e in k4) / 1e6:.2f} ms, "
      f"union {union(k4) / 1e6:.2f} ms ({100 * union(k4) / span:.1f} % of the window), mean duration "
      f"{sum(e - s for s, e in k4) / len(k4) / 1e3:.1f} us")
print(f"other kernels: {len(oth)} launches, sum {sum(e - s for s, e in oth) / 1e6:.2f} ms, union {union(oth) / 1e6:.2f} ms; "
      f"any kernel: union {union(k4 + oth) / 1e6:.2f} ms ({100 * union(k4 + oth) / span:.1f} %)")
import collections
conc = collections.Counter()
ev = sorted([(s_, 1) for s_, e_ in k4] + [(e_, -1) for s_, e_ in k4])
lvl, last = 0, win0
for t, d in ev:
    conc[lvl] += t - last
    lvl, last = lvl + d, t
print("time share by number of concurrent scoring kernels:", {k: f"{100 * v / span:.1f}%" for k, v in sorted(conc.items())})
print(f"batches per ms in the window: {len(k4) / (span / 1e6):.2f}  -> {32 * len(k4) / (span / 1e9):,.0f} votings/s")
