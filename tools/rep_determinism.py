"""Development aid: bit-for-bit repeatability of the whole path.  For several small shapes (few images: workgroups of a
launch then share CUs only in its tail, the condition of round 2's compaction flake), both scoring modes and every
compaction tiling, one call is repeated `reps` times and records, pixel list, hypotheses, counts, winners and key-points
must equal those of the first call.
    python tools/rep_determinism.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvnet_amd import synth, voting  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
SHAPES = [(3, 200, 280, 31, 700), (1, 480, 640, 40, 512), (2, 120, 160, 14, 96), (5, 240, 320, 45, 300), (8, 480, 640, 40, 1024)]
KEYS = ("rec", "pix", "hyp", "counts", "win")
total_bad = 0
for b, h, w, radius, hn in SHAPES:
    mask, planar, _ = synth.make_batch(b, first_index=1300 + b, h=h, w=w, radius=radius, noise=True, background="normal")
    m = torch.from_numpy(mask).to(dev)
    v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
    for literal in (False, True):
        ref = None
        for kg in ("3", "1", "9"):
            os.environ["PVNET_COMPACT_KG"] = kg
            voting.reload_tuning()
            bad = 0
            n = reps if not literal else max(10, reps // 5)
            for rep in range(n):
                out, d = voting.ransac_voting_layer_v3(m, v, hn, inlier_thresh=0.99, seed=9, literal=literal, return_debug=True)
                tn = [int(x) for x in d["tn"]]
                got = {k: d[k].clone() for k in KEYS}
                got["out"] = out.clone()
                if ref is None:
                    ref = got
                    continue
                same = torch.equal(got["out"], ref["out"]) and all(torch.equal(got[k], ref[k]) for k in ("hyp", "counts", "win"))
                for bi in range(b):
                    same = same and torch.equal(got["rec"][bi, :, :tn[bi]], ref["rec"][bi, :, :tn[bi]]) \
                        and torch.equal(got["pix"][bi, :tn[bi]], ref["pix"][bi, :tn[bi]])
                bad += not same
            total_bad += bad
            print(f"b={b} {h}x{w} hn={hn} {'literal' if literal else 'fast   '} KG {kg}: bad {bad} of {n}", flush=True)
os.environ.pop("PVNET_COMPACT_KG", None)
voting.reload_tuning()
print("TOTAL bad", total_bad)
sys.exit(1 if total_bad else 0)
