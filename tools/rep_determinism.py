"""Development aid: repeat one fast-mode call hundreds of times per compaction tiling (PVNET_COMPACT_KG = 1 / 3 / 9) and
compare the compacted records with a reference run -- the test that exposed (and now guards) the K2 LDS flake.
    python tools/rep_determinism.py [reps]"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvnet_amd import synth, voting
dev = torch.device("cuda:0")
mask, planar, _ = synth.make_batch(3, first_index=1300, h=200, w=280, radius=31, noise=True, background="normal")
m = torch.from_numpy(mask).to(dev); v = synth.planar_to_vertex_view(torch.from_numpy(planar).to(dev))
def fast():
    of, df = voting.ransac_voting_layer_v3(m, v, 700, inlier_thresh=0.99, seed=9, return_debug=True)
    return df["rec"].clone(), df["tn"].clone()
os.environ["PVNET_COMPACT_KG"] = "3"; voting.reload_tuning()
ref, tn = fast()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for kg in ("1", "3", "9", "1"):
    os.environ["PVNET_COMPACT_KG"] = kg; voting.reload_tuning()
    bad = 0
    for rep in range(n):
        got, _ = fast()
        ok = all(torch.equal(ref[bi, :, :int(tn[bi])], got[bi, :, :int(tn[bi])]) for bi in range(3))
        bad += not ok
    print(os.environ.get("PVNET_VOTE_LIB", "default").split("/")[-1], "serialize", os.environ.get("AMD_SERIALIZE_KERNEL"), "KG", kg, "bad", bad, "of", n, flush=True)
