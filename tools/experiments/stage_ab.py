"""Same-box A/B of library builds by STAGE (development aid, round 6): for every name=path[,ENV=value,...] the hypotheses and scoring
stages (20 back-to-back launches each, events) and the whole call on one stream (VotePlan, 200 calls), each in a subprocess with
PVNET_VOTE_LIB pointing at the build; the benchmark's noisy field and, with `clean`, the clean one.
    python tools/experiments/stage_ab.py [clean] [--rounds 2] name=path[,ENV=value] ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("STAGE_AB_CHILD"):
    import time

    import torch
    sys.path.insert(0, ROOT)
    from pvnet_amd import synth, voting
    dev = torch.device("cuda:0")
    out = []
    for clean in ([False, True] if "clean" in sys.argv else [False]):
        mask, planar, _ = synth.make_batch(32, radius=40, noise=not clean, background="zeros" if clean else "normal")
        m = torch.from_numpy(mask).to(dev)
        v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
        st = {s: min(voting.stage_repeat_ms(m, v, 1024, inlier_thresh=0.99, stage=s, repeats=20, both=True)[1] for _ in range(3)) * 1e3
              for s in ("hypotheses", "score", "select_refine")}
        plan = voting.VotePlan(m, v, 1024, inlier_thresh=0.99)
        for _ in range(50):
            plan(m, v, seed=1)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            for i in range(200):
                plan(m, v, seed=i)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 200 * 1e6)
        out.append(f"{'clean' if clean else 'noisy'}: hyp {st['hypotheses']:5.1f}  score {st['score']:6.1f}  refine {st['select_refine']:5.1f}  call {best:6.1f} us")
    print(" | ".join(out))
    sys.exit(0)
args = [a for a in sys.argv[1:]]
rounds = 2
if "--rounds" in args:
    i = args.index("--rounds"); rounds = int(args[i + 1]); del args[i:i + 2]
libs = [a for a in args if "=" in a]
rest = [a for a in args if "=" not in a]
for r in range(rounds):
    for spec in libs:
        n, p = spec.split("=", 1)
        p, *envs = p.split(",")
        env = dict(os.environ, STAGE_AB_CHILD="1", PVNET_VOTE_LIB=os.path.abspath(p), **dict(e.split("=", 1) for e in envs))
        o = subprocess.run([sys.executable, os.path.abspath(__file__)] + rest, env=env, capture_output=True, text=True)
        print(f"round {r} {n:10s} {o.stdout.strip().splitlines()[-1] if o.stdout.strip() else 'FAILED ' + o.stderr[-400:]}", flush=True)
