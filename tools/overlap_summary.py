"""Round 6 (VERDICT r05 "Next" 3): what the HEADLINE measures, from a rocprofv3 --kernel-trace database of the six-stream run.

    python tools/overlap_summary.py <trace_results.db> [out.txt] [csv] [focus]

`focus` (default "score_exact_kernel_both_0_1": the merged dense + disc-culling launch with contiguous runs in its dense body, the scoring kernel that only calls flagged PVNET_F_CONCURRENT run):
the window is the middle 80 % of the longest stretch of dispatches of THAT kernel no other scoring kernel interrupts -- bench.py's
six-stream regions -- so the
single-stream and approximate-mode regions of the same run stay out of the numbers.

Prints (and writes): per-kernel durations under concurrency; the share of wall time with 0 / 1 / >= 2 scoring kernels resident;
the steps of every stream taken apart -- a step is the stream's five consecutive launches mask, compact, hypotheses, score,
select/refine: its span, the sum of its kernels, the gaps between its dependent launches, the gap to the stream's next step -- and
how many steps complete per second in the window.  The optional csv keeps (name, start, end, group) of every dispatch of the
window so that the summary can be re-derived without the database (which is too large to merge back)."""
import collections
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"(pvd::)?\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n.strip()


def stage_of(n):
    for key, st in (("mask_bits", "K1 mask"), ("compact", "K2 compact"), ("hypothesis", "K3 hypotheses"), ("score", "K4 score"),
                    ("select_refine", "K5 select/refine")):
        if key in n:
            return st
    return None


def union_len(iv):
    tot, cs, ce = 0, None, None
    for s, e in sorted(iv):
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    return tot + (ce - cs if cs is not None else 0)


def level_shares(iv, w0, w1):
    ev = sorted([(max(s, w0), 1) for s, e in iv] + [(min(e, w1), -1) for s, e in iv])
    lvl, last, out = 0, w0, collections.Counter()
    for t, d in ev:
        out[lvl] += t - last
        lvl, last = lvl + d, t
    out[lvl] += w1 - last
    return out


def main(db, out=None, csv=None, focus="score_exact_kernel_both_0_1", f0=0.1, f1=0.9):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    grp = next((c for c in ("stream_id", "stream", "queue_id", "queue") if c in cols), None)
    rows = cur.execute(f"select name, start, end, {grp or '0'} from kernels order by start").fetchall()
    rows = [(short(n), s, e, g) for n, s, e, g in rows if stage_of(n)]
    # the longest stretch of `focus` dispatches that no OTHER scoring kernel interrupts (bench.py's single-stream / approximate-mode
    # regions and its per-stage profile launch other variants)
    best, cur_run = [], []
    for r in rows:
        if stage_of(r[0]) != "K4 score":
            continue
        if focus in r[0]:
            cur_run.append(r)
        else:
            best, cur_run = (cur_run if len(cur_run) > len(best) else best), []
    foc = (cur_run if len(cur_run) > len(best) else best) or rows
    t0, t1 = min(r[1] for r in foc), max(r[2] for r in foc)
    w0, w1 = t0 + f0 * (t1 - t0), t0 + f1 * (t1 - t0)
    win = [r for r in rows if r[1] >= w0 and r[2] <= w1]
    span = w1 - w0
    L = []
    P = L.append
    P(f"source: {db}  (`kernels` grouped by `{grp}`); window = {100 * f0:.0f} %..{100 * f1:.0f} % of the span of `{focus}` dispatches = "
      f"{span / 1e6:.2f} ms, {len(win)} dispatches of the path, {len({r[3] for r in win})} {grp or 'group'}s")
    P("")
    P("== per-kernel durations UNDER CONCURRENCY (all dispatches of the window)")
    P("%-46s %6s %10s %10s %10s %10s" % ("kernel", "calls", "avg_us", "p50_us", "min_us", "max_us"))
    byname = collections.defaultdict(list)
    for n, s, e, g in win:
        byname[n].append((e - s) / 1e3)
    for n, d in sorted(byname.items(), key=lambda kv: -sum(kv[1])):
        d = sorted(d)
        P("%-46s %6d %10.2f %10.2f %10.2f %10.2f" % (n[:46], len(d), sum(d) / len(d), d[len(d) // 2], d[0], d[-1]))
    P("")
    k4 = [(s, e) for n, s, e, g in win if stage_of(n) == "K4 score"]
    small = [(s, e) for n, s, e, g in win if stage_of(n) != "K4 score"]
    sh = level_shares(k4, w0, w1)
    P("== share of wall time by the number of scoring kernels resident at once")
    P("   " + "   ".join(f"{k}: {100 * v / span:.1f} %" for k, v in sorted(sh.items())))
    ge2 = sum(v for k, v in sh.items() if k >= 2)
    P(f"   0: {100 * sh.get(0, 0) / span:.1f} %   1: {100 * sh.get(1, 0) / span:.1f} %   >= 2: {100 * ge2 / span:.1f} %")
    P(f"   scoring kernels: sum of durations {sum(e - s for s, e in k4) / 1e6:.2f} ms = {sum(e - s for s, e in k4) / span:.2f} x the "
      f"window; union {100 * union_len(k4) / span:.1f} %; small stages: sum {sum(e - s for s, e in small) / 1e6:.2f} ms = "
      f"{sum(e - s for s, e in small) / span:.2f} x, union {100 * union_len(small) / span:.1f} %; any kernel of the path resident "
      f"{100 * union_len(k4 + small) / span:.1f} %")
    # time with NO scoring kernel resident: which small stages run then
    P("")
    # ---- steps per stream
    order = ["K1 mask", "K2 compact", "K3 hypotheses", "K4 score", "K5 select/refine"]
    steps, per = [], collections.defaultdict(list)
    for n, s, e, g in win:
        per[g].append((s, e, stage_of(n)))
    for g, lst in per.items():
        lst.sort()
        i = 0
        while i + 5 <= len(lst):
            if [x[2] for x in lst[i:i + 5]] == order:
                steps.append((g, lst[i:i + 5]))
                i += 5
            else:
                i += 1
    if steps:
        P(f"== the critical path of one step ({len(steps)} complete steps in the window; a step = the five consecutive launches of one {grp})")
        dur = collections.defaultdict(list)
        gap = collections.defaultdict(list)
        spans, sums = [], []
        for g, st in steps:
            spans.append((st[4][1] - st[0][0]) / 1e3)
            sums.append(sum(e - s for s, e, _ in st) / 1e3)
            for j, (s, e, name) in enumerate(st):
                dur[name].append((e - s) / 1e3)
                if j:
                    gap[f"{order[j - 1][:2]} -> {order[j][:2]}"].append((s - st[j - 1][1]) / 1e3)
        med = lambda v: sorted(v)[len(v) // 2]
        P("   stage durations inside a step (us):   " + "   ".join(f"{k}: avg {sum(v) / len(v):.1f} / p50 {med(v):.1f}" for k, v in dur.items()))
        P("   gaps between dependent launches (us): " + "   ".join(f"{k}: avg {sum(v) / len(v):.2f} / p50 {med(v):.2f}" for k, v in gap.items()))
        P(f"   step span K1 start -> K5 end: avg {sum(spans) / len(spans):.1f} us, p50 {med(spans):.1f}; sum of its five kernels avg "
          f"{sum(sums) / len(sums):.1f} us; so dependent-launch gaps cost {sum(spans) / len(spans) - sum(sums) / len(sums):.1f} us per step")
        nxt = []
        bygrp = collections.defaultdict(list)
        for g, st in steps:
            bygrp[g].append(st)
        for g, sl in bygrp.items():
            for a, b in zip(sl, sl[1:]):
                nxt.append((b[0][0] - a[4][1]) / 1e3)
        if nxt:
            P(f"   gap between a {grp}'s consecutive steps (K5 end -> next K1 start): avg {sum(nxt) / len(nxt):.2f} us, p50 {med(nxt):.2f}")
        rate = len(steps) / (span / 1e9)
        P(f"   steps completed in the window: {len(steps)} in {span / 1e6:.2f} ms = {span / 1e3 / len(steps):.1f} us per step "
          f"(under the profiler) -> {32 * rate:,.0f} votings/s at batch 32")
        P(f"   = per step: scoring kernel avg {sum(dur['K4 score']) / len(dur['K4 score']):.1f} us issued {len(k4)} times in {span / 1e3:.0f} us: "
          f"scoring-kernel residency per step {sum(e - s for s, e in k4) / 1e3 / len(steps):.1f} us, wall per step {span / 1e3 / len(steps):.1f} us")
    txt = "\n".join(L) + "\n"
    print(txt)
    if out:
        open(out, "w").write(txt)
    if csv:
        with open(csv, "w") as f:
            f.write("name,start_ns,end_ns,group\n")
            for n, s, e, g in win:
                f.write(f"{n},{s - t0},{e - t0},{g}\n")


if __name__ == "__main__":
    main(*sys.argv[1:5])
