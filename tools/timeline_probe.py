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

# ---- which kernels run while NO scoring kernel is resident (= the exposed part of a step), per batch
import re
k4s = sorted(k4)
gaps, cs, ce = [], None, None
for s_, e_ in k4s:  # complement of the union of the scoring launches inside the window
    if cs is None:
        cs, ce = s_, e_
    elif s_ <= ce:
        ce = max(ce, e_)
    else:
        gaps.append((ce, s_))
        cs, ce = s_, e_
gap_total = sum(b - a for a, b in gaps)
nb = max(1, len(k4))
print(f"no scoring kernel running: {gap_total / 1e6:.2f} ms of the window = {gap_total / nb / 1e3:.1f} us per batch ({len(gaps)} gaps)")
gl = sorted(b - a for a, b in gaps)
if gl:
    print("gap lengths (us) percentiles 10 / 50 / 90 / max:", [round(gl[min(len(gl) - 1, int(len(gl) * q))] / 1e3, 1) for q in (0.1, 0.5, 0.9)], round(gl[-1] / 1e3, 1),
          "; gaps > 50 us:", sum(1 for g in gl if g > 50e3), "holding", round(sum(g for g in gl if g > 50e3) / 1e6, 2), "ms")
    big = [(a, b) for a, b in gaps if b - a > 50e3]
    if len(big) > 2:
        d = sorted((big[i + 1][0] - big[i][0]) / 1e6 for i in range(len(big) - 1))
        print("distance between the starts of consecutive long gaps (ms): median", round(d[len(d) // 2], 2), "min", round(d[0], 2), "max", round(d[-1], 2))


def short(n):
    m = re.search(r"(\w+_kernel)", n)
    return m.group(1) if m else n[:40]


alone, total, cnt = collections.Counter(), collections.Counter(), collections.Counter()
gi = 0
for n, s_, e_ in sorted((r for r in rows if "score" not in r[0]), key=lambda r: r[1]):
    total[short(n)] += e_ - s_
    cnt[short(n)] += 1
    for a, b in gaps:  # (few hundred gaps: a linear scan is fine)
        if b <= s_:
            continue
        if a >= e_:
            break
        alone[short(n)] += min(e_, b) - max(s_, a)
print("per batch, microseconds: kernel  duration  of which while no scoring kernel runs")
for n in sorted(total, key=lambda k: -total[k]):
    print(f"  {n:28s} {total[n] / nb / 1e3:7.1f}  {alone[n] / nb / 1e3:7.1f}   (launches per batch {cnt[n] / nb:.2f}, mean {total[n] / cnt[n] / 1e3:.1f} us)")
# scoring launches by duration: are launches stretched when they share the chip?
durs = sorted(e_ - s_ for s_, e_ in k4)
print("scoring launch duration percentiles 10 / 50 / 90 (us):", [round(durs[int(len(durs) * q)] / 1e3, 1) for q in (0.1, 0.5, 0.9)])
