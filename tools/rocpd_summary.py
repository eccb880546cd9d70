"""Turns rocprofv3's rocpd sqlite outputs (gpurun_out/prof_<tag>/) into the small text summaries kept under
profiles/.   python tools/rocpd_summary.py gpurun_out/prof_r01a profiles/r01a"""
import glob
import os
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"(pvd::)?\(anonymous namespace\)::", "", name)
    name = re.sub(r"\((pvd::)?VoteParams\)", "", name)
    return name[:90]


def inst(name):
    """the kernel's name WITH its template arguments ("score_exact_kernel<8, false, false>"): the json summaries are keyed
    by instantiation, so that the literal-mode instantiations a profiled run also launches are never mixed into the
    timed ones"""
    name = re.sub(r"(pvd::)?\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\s*\[clone[^\]]*\]\s*$", "", name)
    name = re.sub(r"\(.*\)$", "", name).strip()
    # round 4: the exact scoring kernel's instantiations are plain functions (one amdgpu_num_vgpr literal each,
    # k4_score_exact.hip PV_DEF_SCORE_EXACT): score_exact_kernel_<MH>_<FOLD>_<TIMED>_<NACC>_<RUNS> -> the template spelling
    m = re.match(r"(score_exact_kernel)_(\d+)_(\d+)_(\d+)_(\d+)_(\d+)$", name)
    if m:
        b = lambda x: "true" if x != "0" else "false"
        name = f"{m.group(1)}<{m.group(2)}, {m.group(3)}, {b(m.group(4))}, {m.group(5)}, {b(m.group(6))}>"
    # round 6: the merged dense + disc-culling launch, score_exact_kernel_both_<TIMED>_<RUNS> (k4_score_cull.hip): the dense body is
    # score_exact_body<8, 1, TIMED, 1, RUNS>
    m = re.match(r"score_exact_kernel_both_(\d+)_(\d+)$", name)
    if m:
        b = lambda x: "true" if x != "0" else "false"
        name = f"score_exact_kernel<8, 1, {b(m.group(1))}, 1, {b(m.group(2))}, +culling body>"
    return name if re.match(r"\w+_kernel(<.*>)?$", name) else None


def source_hash():
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    return bench.source_hash()


def kernel_stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    out = ["%-92s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%")]
    for n, c, s, a, mn, mx in rows:
        out.append("%-92s %7d %12.1f %10.2f %10.2f %10.2f %6.2f" % (short(n), c, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3,
                                                                    100 * s / tot))
    return "\n".join(out)


def pmc(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                       "group by kernel_name, counter_name order by kernel_name").fetchall()
    out = ["%-92s %-22s %6s %16s %12s" % ("kernel", "counter", "n", "avg_value", "avg_dur_us")]
    for k, c, n, v, d in rows:
        if "at::native" in k or "rocclr" in k:
            continue
        out.append("%-92s %-22s %6d %16.2f %12.2f" % (short(k), c, n, v, d / 1e3))
    return "\n".join(out)


def main(src, dst):
    parts = []
    for db in sorted(glob.glob(os.path.join(src, "trace*", "*.db"))):
        parts.append(f"== rocprofv3 --kernel-trace --stats  ({os.path.relpath(db, src)}) : per-kernel durations\n" +
                     kernel_stats(db))
    for db in sorted(glob.glob(os.path.join(src, "pmc*", "*.db"))):
        parts.append(f"== rocprofv3 --pmc pass ({os.path.relpath(db, src)}) : average counter value per dispatch\n" +
                     "   (FETCH_SIZE / WRITE_SIZE are KB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x\n"
                     "    -- MI355X_MICROARCH.md 'HBM'; profiled passes run at lower clocks than un-profiled ones)\n" +
                     pmc(db))
    # per-launch HBM traffic of every kernel of the path (for bench.py's roofline.traffic)
    traffic = {}
    for kind in ("fetch", "write"):
        for db in glob.glob(os.path.join(src, f"pmc_{kind}", "*.db")):
            cur = sqlite3.connect(db).cursor()
            for k, v, n in cur.execute("select kernel_name, avg(value), count(*) from counters_collection where "
                                       "counter_name=? group by kernel_name", (kind.upper() + "_SIZE",)):
                name = inst(k)
                if name:
                    traffic.setdefault(name, {})[kind + "_kb"] = v
                    traffic[name]["n"] = n
    if traffic:
        import json
        with open(dst + "_traffic.json", "w") as f:
            json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), average per dispatch, KB; keyed by "
                                 "template instantiation, n = dispatches",
                       "source_hash": source_hash(),
                       "note": "gfx950 FETCH_SIZE counts wide coalesced reads at 1/2 (MI355X_MICROARCH.md, HBM): "
                               "bench.py doubles fetch_kb",
                       "kernels": traffic}, f, indent=1)
    # every counter of every --pmc pass, averaged per dispatch, keyed by kernel (for bench.py's roofline.mfma_util)
    counters = {}
    for db in sorted(glob.glob(os.path.join(src, "pmc*", "*.db"))):
        cur = sqlite3.connect(db).cursor()
        for k, c, v, d, n in cur.execute("select kernel_name, counter_name, avg(value), avg(duration), count(*) from "
                                         "counters_collection group by kernel_name, counter_name"):
            name = inst(k)
            if name:
                counters.setdefault(name, {})[c] = v
                counters[name].setdefault("avg_duration_us", {})[c] = d / 1e3
                counters[name]["n"] = n
    if counters:
        import json
        with open(dst + "_pmc.json", "w") as f:
            json.dump({"source": "rocprofv3 --pmc passes (own runs, --kernel-trace only), average per dispatch; keyed by template "
                                 "instantiation, n = dispatches",
                       "source_hash": source_hash(),
                       "note": "SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the chip's SIMDs; GRBM_GUI_ACTIVE is "
                               "summed over the 8 XCDs: MfmaUtil = MFMA_BUSY / (1024 SIMDs x GUI_ACTIVE / 8)",
                       "kernels": counters}, f, indent=1)
    txt = "\n\n".join(parts) + "\n"
    with open(dst + "_rocprof_summary.txt", "w") as f:
        f.write(txt)
    for j in ("bench.json", "trace_bench.json"):
        p = os.path.join(src, j)
        if os.path.exists(p):
            with open(p) as f, open(dst + "_" + j, "w") as g:
                g.write(f.read())
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
