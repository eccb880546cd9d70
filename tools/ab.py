"""Same-box A/B of library builds (boxes of the pool differ by +-5 %, so every comparison is made inside ONE gpurun call):
    python tools/ab.py [--rounds 2] name=path/to/lib.so[,ENV=value,...] [name=path ...] [bench.py options]
runs bench.py (exact mode, no CPU baseline, no parity block) with PVNET_VOTE_LIB pointing at each library in turn, `rounds`
times interleaved, and prints the six-stream rate, the single-stream rate and the scoring kernel's own duration."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
rounds, extra = 2, []
libs = []
i = 0
while i < len(args):
    if args[i] == "--rounds":
        rounds = int(args[i + 1]); i += 2
    elif "=" in args[i] and not args[i].startswith("--"):
        n, p = args[i].split("=", 1)          # name=path[,ENV=value,...]
        p, *envs = p.split(",")
        libs.append((n, (os.path.abspath(p), dict(e.split("=", 1) for e in envs)))); i += 1
    else:
        extra.append(args[i]); i += 1
res = {n: [] for n, _ in libs}
for r in range(rounds):
    for n, p in libs:
        env = dict(os.environ, PVNET_VOTE_LIB=p[0], **p[1])
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-parity", "--regions", "7"] + extra,
                             env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            row = (d["value"], d["single_stream"]["value"], d["roofline"]["avg_launch_ms"] * 1e3, d["regions"]["spread"])
        except Exception as e:  # noqa: BLE001
            print(n, "FAILED", e, out.stderr[-500:])
            continue
        res[n].append(row)
        print(f"round {r} {n:12s} six-stream {row[0] / 1e3:7.1f} k   single {row[1] / 1e3:7.1f} k   score kernel {row[2]:6.1f} us   spread {row[3] * 100:.1f} %", flush=True)
print()
for n, _ in libs:
    if res[n]:
        m = [sum(x[j] for x in res[n]) / len(res[n]) for j in range(3)]
        print(f"MEAN    {n:12s} six-stream {m[0] / 1e3:7.1f} k   single {m[1] / 1e3:7.1f} k   score kernel {m[2]:6.1f} us")
